"""Replay harness — NOT product code.

C++ transliteration of the reference's host-side control layer (TrackerContext, SelectionState,
TimingStats: /root/reference/src/tracker_context.rs, selection_state.rs, timing_stats.rs) with the
same names and semantics, exported through include/vittrack_host.h and driven from the tests
(SURVEY.md §8 f1). It exists so that the probe-closure call sequence of the reference
(src/pipeline.rs:67-184) can be replayed against libvittrack_hip.so; the Rust host keeps its own
control layer, and libvittrack_hip.so neither links nor loads anything in this directory.
"""
