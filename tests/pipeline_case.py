"""The timed path of bench.py under the oracle, run by tests/test_pipeline_gpu.py in its own process (torch first, then the HIP
library):  python pipeline_case.py <config> <frames> [rotations]

A resident batch goes through orb_slam2_aruco_amd.pipeline.FrontEndPipeline -- the class bench.py times: extract_batch_device
+ aruco detect_batch_device + marker poses + orbfe_knn2_batch_device + orbfe_search_for_initialization_batch_device with
bench.py's arguments -- and EVERY frame's records and EVERY pair's best_idx / best_dist / second_dist / matches12 / nmatches are
compared with the oracle (ORBmatcher.cc:409-524 for the windowed pass).  Several steps run back to back on rotated copies of
the stream, as in the bench, so the result sets in rotation, the cross-step stream dependencies and the halo slot (the pair across the batch boundary) are
exercised too."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import numpy as np
import torch
torch.cuda.init()
from orb_slam2_aruco_amd import synth, sharding
from orb_slam2_aruco_amd.pipeline import FrontEndPipeline
import oracle_lib as oracle
import pipeline_check
sys.path.insert(0, os.path.dirname(HERE))
import bench

cfg = dict(bench.CONFIGS[sys.argv[1]])
B = int(sys.argv[2])
R = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rows, cols = cfg["rows"], cfg["cols"]
frames = synth.stream(rows, cols, B, sharding.stream_seed(0), cfg["dictionary"], n_markers=cfg["n_markers"])
pipe = FrontEndPipeline(B, rows, cols, cfg["nfeatures"], cfg["nlevels"], cfg["dictionary"], device=0)
shifts = [(r * B) // R for r in range(R)]
batches = [pipe.upload(np.roll(frames, -s, axis=0)) for s in shifts]
pipe.warmup(batches[0], 1)
# three back-to-back steps without a sync in between; the last one decides what is read back
order = [1 % R, 0, (R - 1)]
for r in order:
    cur = pipe.step(batches[r])
pipe.synchronize()
assert not any(pipe.status().values()), pipe.status()
rec = pipe.read_records(cur)
matches = pipe.read_matches()
host = np.roll(frames, -shifts[order[-1]], axis=0)
host_prev = np.roll(frames, -shifts[order[-2]], axis=0)
# every pair inside the batch, and the pair across the batch boundary: the previous step's last frame against this step's first
res = pipeline_check.check_against_oracle(oracle, host, list(range(B)), rec, matches, cfg["nfeatures"], cfg["nlevels"], cfg["dictionary"],
                                          cols, rows, pipe.cam_K, pipe.cam_D, pairs=list(range(B - 1)), prev_last=host_prev[B - 1])
assert res["pairs_checked"] == B - 1 and res["boundary_pair_checked"] and res["keypoints_checked"] > 100 * B and res["markers_checked"] > 0, res
# the other result set still holds the step before: spot-check it (double buffering must not have been overwritten)
rec_prev = pipe.read_records((cur - 1) % pipe.R)
pipeline_check.check_against_oracle(oracle, host_prev, [0, B - 1], rec_prev, None, cfg["nfeatures"], cfg["nlevels"], cfg["dictionary"],
                                    cols, rows, pipe.cam_K, pipe.cam_D)
print("ok", res["keypoints_checked"], res["markers_checked"], res["pairs_checked"])
