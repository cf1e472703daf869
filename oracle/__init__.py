"""CPU oracle for the tracker hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may import this
package; nothing under gstreamer-vit-tracker_amd/ does. See oracle/README.md for what each part
restates and its parity status (the model stages are PARITY UNPINNED against the reference: the
reference's tracker lives in an un-vendored crate and ships no tests).
"""
