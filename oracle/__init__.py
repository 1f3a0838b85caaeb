"""CPU oracle for the MI355X-native DeepDenoiser hot path.  TEST INFRASTRUCTURE ONLY:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
PARITY UNPINNED against live TensorFlow (see oracle/tf_ops.py and DESIGN.md)."""
