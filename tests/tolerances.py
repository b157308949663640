"""Tolerances of the MULTI-STEP parity tests (losses / last embeddings / final weights after the fixtures' recorded steps), stated as a
multiple of a measured quantity instead of asserted (VERDICT round 2, weak spot 1a).

SPREAD_* is the largest difference, over the fixtures, between equally valid CPU executions of the reference's own op sequence
(oracle/torch_cpu_path.py, i.e. torch.sparse.mm / nn.Linear / F.elu / F.normalize / Adam): fp32 with 1 thread, fp32 with 8 threads and
fp64, all from the same start on the same recorded batches.  tests/test_trajectory_spread.py re-measures it on every run of the CPU
suite and fails if it ever exceeds the recorded figure, so the tolerances below stay tied to a number that is shown, not claimed.
Measured in this container (torch 2.10 CPU): losses 1.4e-7 relative, embeddings 1.7e-6 of the largest entry, weights 1.7e-3 lr -- the
trajectory is NOT sensitive at the level of whole learning-rate steps (an earlier DESIGN.md said it was; it is not).

A device run differs from the fixture by its own fp32 rounding (different summation order in the SpMM segments, MFMA k order, the
fixed-order weight-gradient reduce); tools/trajectory_deviation.py prints what it measures on the GPU box
(profiles/r03_trajectory_deviation.txt).  TRAJ_K x the CPU spread bounds it with room for a different box."""

SPREAD_LOSS_REL = 1.5e-7       # max |loss_a - loss_b| / max |loss|
SPREAD_EMB_REL = 2.0e-6        # max |emb_a - emb_b| / max |emb|
SPREAD_WEIGHT_LR = 2.0e-3      # max |w_a - w_b| / lr

TRAJ_K = 10                    # device-vs-fixture tolerances = TRAJ_K x the spread between CPU executions
TRAJ_LOSS_RTOL = TRAJ_K * SPREAD_LOSS_REL      # 1.5e-6   (measured on the device: <= 7.4e-8)
TRAJ_EMB_REL = TRAJ_K * SPREAD_EMB_REL         # 2e-5     (measured: <= 1.3e-6)
TRAJ_WEIGHT_LR = TRAJ_K * SPREAD_WEIGHT_LR     # 0.02 lr  (measured: <= 1.2e-3 lr)
# end to end through the CLI (train.py: kNN graph, beta percentile, 12 steps, '%.18e' text): absolute difference of unit-norm rows
TRAJ_CLI_EMB_ABS = 2e-5
