"""Report files in the reference's text formats, so that a run on the GPU can be diffed against a ReaK run elsewhere.

Mirrors (host side only, no device work):
  * vlist_sbmp_report::draw_motion_graph  (R/ctrl/path_planning/vlist_sbmp_report.hpp:102-112): one file
    "<path>vlist_<num_vertices, 6 digits zero-filled>" with one line per vertex in vertices(g) order;
  * any_mg_vertex_printer::operator()      (R/ctrl/path_planning/any_motion_graphs.hpp:666-700): per line, every
    coordinate as " " + setw(10) in the stream's default float format, then distance_accum (optimal graphs), then
    density (dense graphs), then std::endl;
  * timing_sbmp_report::draw_motion_graph  (basic_sbmp_reporters.hpp:350-354): "<num_vertices> <microseconds>";
  * least_cost_sbmp_report                 (basic_sbmp_reporters.hpp:532-580): progress "<num_vertices> <best>",
    solutions "<last_node_count> <best>", best starts at 1e10.
The stream format is the C++ default (precision 6, %g); tests/test_reports.py pins it against a real iostream.
"""
import os

import numpy as np

BASIC_MOTION_GRAPH_KIND = 0x00
OPTIMAL_MOTION_GRAPH_KIND = 0x01
DENSE_MOTION_GRAPH_KIND = 0x10


def cxx_double(x):
    """operator<<(std::ostream&, double) with default flags: printf("%g") at precision 6."""
    return "%g" % float(x)


def _field(x):
    return " " + cxx_double(x).rjust(10)


def vlist_text(pos, distance_accum=None, density=None, graph_kind=None):
    pos = np.asarray(pos, dtype=np.float64)
    if graph_kind is None:
        graph_kind = (OPTIMAL_MOTION_GRAPH_KIND if distance_accum is not None else 0) | \
                     (DENSE_MOTION_GRAPH_KIND if density is not None else 0)
    if (graph_kind & OPTIMAL_MOTION_GRAPH_KIND) and distance_accum is None:
        raise ValueError("optimal motion-graph kind needs vertex_distance_accum")
    if (graph_kind & DENSE_MOTION_GRAPH_KIND) and density is None:
        raise ValueError("dense motion-graph kind needs vertex_density")
    out = []
    for v in range(pos.shape[0]):
        line = "".join(_field(c) for c in pos[v])
        if graph_kind & OPTIMAL_MOTION_GRAPH_KIND:
            line += _field(distance_accum[v])
        if graph_kind & DENSE_MOTION_GRAPH_KIND:
            line += _field(density[v])
        out.append(line + "\n")
    return "".join(out)


def vlist_file_name(file_path, num_vertices):
    return "%svlist_%06d" % (file_path, num_vertices)


def write_vlist(file_path, pos, distance_accum=None, density=None, graph_kind=None):
    """One draw_motion_graph call of vlist_sbmp_report; returns the file written."""
    name = vlist_file_name(file_path, len(pos))
    with open(name, "w") as f:
        f.write(vlist_text(pos, distance_accum, density, graph_kind))
    return name


def write_rrt_progress(file_path, pos, progress_interval):
    """The vlist files a plain RRT run leaves behind: report_progress fires on every progress_interval-th
    vertex_added (motion_planner_base.hpp:343-347; the root counts), and RRT vertices never change once added,
    so the snapshot at n vertices is the first n rows."""
    names = []
    n = len(pos)
    for m in range(progress_interval, n + 1, progress_interval):
        names.append(write_vlist(file_path, pos[:m]))
    return names


def timing_line(num_vertices, microseconds):
    return "%d %d\n" % (num_vertices, microseconds)


class LeastCostReport:
    """least_cost_sbmp_report: call progress(n) / solution(cost) in event order; text accumulates in .out / .sol."""

    def __init__(self):
        self.current_best, self.last_node_count = 1e10, 0
        self.out, self.sol = [], []

    def progress(self, num_vertices):
        self.last_node_count = num_vertices
        self.out.append("%d %s\n" % (num_vertices, cxx_double(self.current_best)))

    def solution(self, total_cost):
        if total_cost < self.current_best:
            self.current_best = total_cost
        self.sol.append("%d %s\n" % (self.last_node_count, cxx_double(self.current_best)))


def write_solution_path(name, pos, path):
    """Way-points of a registered solution (start → goal), one printer line per way-point."""
    with open(name, "w") as f:
        f.write(vlist_text(np.asarray(pos)[np.asarray(path, dtype=np.int64)]))
    return os.path.abspath(name)
