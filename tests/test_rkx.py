"""`.rkx` scenario I/O (reak_amd/rkx.py): ReaK XML archives of scenes -- shapes, poses, KTE chain ops -- written and read
back bit-identically, and the obstacle courses of R/examples/misc/build_X8_obstacle_courses.cpp restated as generators.
CPU tests; the GPU run over a course is in test_gpu_parity.py."""
import numpy as np
import pytest

from reak_amd import rkx, scenarios
from reak_amd import types as T


def _same_scene(a, b):
    robot_then_env = [s for s in a.shapes if s.anchor >= 0] + [s for s in a.shapes if s.anchor < 0]
    return (bytes(a.ops_array()) == bytes(b.ops_array()) and bytes(a.base) == bytes(b.base) and
            bytes(T.as_array(robot_then_env, T.Shape)) == bytes(b.shapes_array()) and np.array_equal(a.start, b.start) and
            np.array_equal(a.goal, b.goal) and a.n_dof == b.n_dof and a.n_frames == b.n_frames)


@pytest.mark.parametrize("make", [lambda: scenarios.make_c2(world_seed=1), lambda: scenarios.make_c1(world_seed=2),
                                  lambda: scenarios.make_c3(world_seed=3), lambda: scenarios.make_random_chain(4, seed=2),
                                  lambda: scenarios.make_random_chain(7, seed=1), lambda: scenarios.make_hidim(6)])
def test_scene_round_trip_is_bit_identical(make):
    """scene -> .rkx -> scene: every op of the chain (kinds, frame / coordinate wiring, axes, offsets, masses, tensors),
    the base frame with its gravity, every shape (kind, anchor frame, pose, dimensions), start and goal come back with the
    same bits (doubles are written with 17 significant digits, not ReaK's stream default of 6)."""
    scn = make()
    text = rkx.write_scene(scn)
    assert text.startswith('<?xml version="1.0" encoding="UTF-8" standalone="yes" ?>\n<!DOCTYPE reak_serialization>\n'
                           '<reak_serialization version="2">\n') and text.endswith("</reak_serialization>\n")
    back = rkx.read_scene(text, scn)
    assert _same_scene(scn, back)
    assert rkx.write_scene(back) == text  # and the archive of the copy is the same text


def test_archive_shares_objects_by_id_like_xml_oarchive():
    """A frame referred to by a joint, the next link and a shape is written ONCE; later uses carry its object_ID only
    (xml_archiver.cpp:520-572).  Type ids are the classes' RK_RTTI ids, written as `id.id.0`."""
    scn = scenarios.make_c1(world_seed=1)
    text = rkx.write_scene(scn)
    assert text.count('type_ID="32.4.0"') > scn.n_frames          # frame_3D<double>: 0x20, double = 4
    assert text.count("<Velocity ") == scn.n_frames               # ... but only n_frames bodies
    assert 'type_ID="3255828484.0"' in text and 'type_ID="3272605715.0"' in text  # revolute_joint_3D 0xC2100004, box 0xC3100013
    assert '<mJacobian type_ID="0" version="0" object_ID="0" is_external="false">' in text


@pytest.mark.parametrize("which,n_shapes", [("one_building", 2), ("window_crossing", 9)])
def test_obstacle_courses_of_the_reference(which, n_shapes):
    """build_X8_obstacle_courses.cpp:35-157 as data: floor plane (rotated by pi about x) + boxes with the poses and
    dimensions written there, start / end positions; through the archive and back unchanged."""
    shapes, names, start, end = rkx.obstacle_course(which)
    assert len(shapes) == n_shapes and names[0] == "floor" and shapes[0].kind == T.SHAPE_PLANE
    assert all(s.kind == T.SHAPE_BOX and s.anchor == -1 for s in shapes[1:])
    text = rkx.write_obstacle_course(which)
    s2, n2, st2, en2 = rkx.read_obstacle_course(text)
    assert bytes(T.as_array(shapes, T.Shape)) == bytes(T.as_array(s2, T.Shape)) and n2 == names
    assert np.array_equal(st2, start) and np.array_equal(en2, end)
    if which == "window_crossing":
        assert list(start) == [0.75, 1.0, -1.0] and list(end) == [9.0, 3.0, -7.0]
        assert list(shapes[3].pose.pos) == [3.0, 4.5, -9.0] and list(shapes[3].dims) == [0.2, 3.0, 2.0]  # wall3
