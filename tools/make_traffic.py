"""profiles/traffic.json from a tools/pmc.py summary: HBM-side bytes per launch = (FETCH_SIZE + WRITE_SIZE) * 1024, per stage
of bench.py (a stage = the launches between two timer marks; per-dispatch means times the launches of a stage).
    python tools/make_traffic.py gpurun_out/pmc_summary.json profiles/traffic.json <tag>"""
import json, sys
src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
s = json.load(open(src))
kb = lambda k: (s[k].get("FETCH_SIZE", 0.0) + s[k].get("WRITE_SIZE", 0.0)) * 1024.0 if k in s else 0.0
steps = min(v["dispatches"] for k, v in s.items() if k in ("k_fast_cells", "k_blur7")) if "k_fast_cells" in s else 1
per_step = lambda k: s[k]["dispatches"] / steps if k in s else 0
stage = {
    "resize": ["k_resize_tab", "k_resize_level"], "fast_cells": ["k_fast_cells"],
    "distribute": ["k_distribute_pyr", "k_distribute", "k_level_offsets"], "blur7": ["k_blur7"],
    "orient_describe": ["k_orient_describe"], "knn2": ["k_knn2_mfma", "k_knn2_tiles", "k_knn2_merge"], "search_init": ["k_search_init"],
    "aruco_threshold": [k for k in s if k.startswith("k_adaptive_threshold")], "aruco_pyramid": ["k_half_area"],
    "aruco_contours": ["k_contours_relay", "k_contours_tail"] + [k for k in s if k.startswith("k_contours_t<")],
    "aruco_decode": ["k_prefilter", "k_decode"], "aruco_finalize": ["k_finalize", "k_marker_poses"],
}
out = {"_note": "HBM-side bytes per launch (300-frame batch) = (FETCH_SIZE + WRITE_SIZE) * 1024 from separate rocprofv3 --pmc passes "
                "(tools/pmc.py; --kernel-trace only), per-dispatch mean x launches per step (profiles/%s_pmc_summary.json), summed over the "
                "kernels of a bench stage (tools/make_traffic.py). FETCH_SIZE counts L2 misses to the fabric. gfx950 note "
                "(MI355X_MICROARCH.md, HBM section): FETCH_SIZE under-reports 16 B/lane streaming reads by 2x; these kernels read bytes / "
                "dwords and k_blur7's WRITE_SIZE matches its exact store volume, so no correction is applied." % tag}
for st, ks in stage.items():
    out[st] = int(round(sum(kb(k) * per_step(k) for k in ks)))
json.dump(out, open(dst, "w"))
print({k: v for k, v in out.items() if k != "_note"})
