"""profiles/traffic_<config>.json and profiles/pmc_stage_<config>.json from a tools/pmc.py run:
    python tools/make_profiles.py gpurun_out/pmc_summary.json gpurun_out/pmc_calibration.json <config> <tag>
traffic: HBM-side bytes per launch of every bench stage = sum over the stage's kernels of (FETCH_SIZE / f_r + WRITE_SIZE / f_w) *
1024 * launches per step, with the correction factors f measured by tools/fetch_calib.hip for the kernel's access width (the
gfx950 FETCH_SIZE under-count, MI355X_MICROARCH.md HBM section).  pmc_stage: VALU issue time (SQ_INSTS_VALU x 4 cycles / 1024 SIMDs
/ 2.4 GHz) and lane utilisation per stage.  bench.py reads both only for the configuration they were measured on."""
import json, os, sys
src, cal, config, tag = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = json.load(open(src))
calib = json.load(open(cal)) if os.path.exists(cal) else {}
# dominant global access width of each kernel (bytes per lane): reads, writes -- from the kernel sources
WIDTH = {"k_orient_describe2": (4, 16), "k_decode_warp": (1, 1), "k_decode_vote": (1, 4), "k_decode_otsu": (16, 4), "k_sfi_rows": (16, 4), "k_sfi_accept": (16, 4),
         "k_sfi_grid": (16, 16), "k_resize_tab": (8, 4), "k_resize_level": (1, 4), "k_fast_cells": (4, 4), "k_blur7": (4, 4), "k_orient_describe": (4, 16),
         "k_adaptive_threshold": (4, 8), "k_half_area4": (8, 4), "k_half_pyr": (16, 8), "k_half_area": (1, 1), "k_knn2_mfma": (16, 4), "k_search_init": (16, 4), "k_contours_relay": (4, 4),
         "k_contours_small": (4, 4), "k_ct_walk": (4, 4), "k_ct_lists": (4, 8), "k_ct_points": (4, 4), "k_tail_prep": (8, 16), "k_tail_approx": (4, 4), "k_tail_finish": (4, 4), "k_decode": (1, 4), "k_distribute_pyr": (4, 4),
         "k_blur7_mfma": (16, 4), "k_threshold_mfma": (16, 4), "k_threshold_pyr": (4, 8), "k_speck_clean": (4, 4)}
names = {1: "unsigned char", 4: "unsigned int", 8: "HIP_vector_type<unsigned int, 2u>", 16: "HIP_vector_type<unsigned int, 4u>"}


def factor(kind, ctr, width):
    for k, v in calib.items():
        if k.startswith("calib_" + kind) and names[width] in k and ctr in v and v[ctr] > 0.1:
            return v[ctr]
    return 1.0


def kb(k):
    if k not in s:
        return 0.0
    base = next((b for b in WIDTH if k.startswith(b)), None)
    wr, ww = WIDTH.get(base, (4, 4))
    return (s[k].get("FETCH_SIZE", 0.0) / factor("read", "FETCH_SIZE", wr) + s[k].get("WRITE_SIZE", 0.0) / factor("write", "WRITE_SIZE", ww)) * 1024.0


def named(*bases):
    """kernels of the summary called `base` or an instantiation `base<...>` of it"""
    return [k for k in s if any(k == b or k.startswith(b + "<") for b in bases)]


steps = min(s[k]["dispatches"] for k in named("k_fast_cells", "k_blur7", "k_blur7_mfma")) if "k_fast_cells" in s else \
    min(s[k]["dispatches"] for k in named("k_adaptive_threshold_t", "k_adaptive_threshold", "k_threshold_pyr", "k_threshold_mfma"))
per_step = lambda k: s[k]["dispatches"] / steps if k in s else 0
stage = {
    "resize": named("k_resize_tab"), "fast_cells": named("k_fast_cells"),
    "distribute": named("k_distribute_pyr", "k_distribute", "k_level_offsets"), "blur7": named("k_blur7", "k_blur7_mfma"),
    "orient_describe": named("k_orient_describe", "k_orient_describe2"), "knn2": named("k_knn2_mfma", "k_knn2_tiles", "k_knn2_merge"),
    "search_init": named("k_search_init", "k_sfi_grid", "k_sfi_rows", "k_sfi_accept"),
    "aruco_threshold": named("k_adaptive_threshold_t", "k_adaptive_threshold", "k_threshold_pyr", "k_threshold_mfma"), "aruco_pyramid": named("k_half_area", "k_half_area4", "k_half_pyr", "k_resize_level"),
    "aruco_contours": [k for k in s if k.startswith("k_contours") or k.startswith("k_tail_") or k.startswith("k_ct_") or k.startswith("k_speck")],
    "aruco_decode": named("k_prefilter", "k_decode", "k_decode_warp", "k_decode_otsu", "k_decode_vote"), "aruco_finalize": named("k_finalize", "k_marker_poses"),
}
traffic = {"_note": "HBM-side bytes per launch (%s batch) = (FETCH_SIZE / f_read + WRITE_SIZE / f_write) * 1024 from separate rocprofv3 --pmc "
                    "passes (tools/pmc.py; --kernel-trace only), per-dispatch mean x launches per step (profiles/%s_pmc_summary.json), summed "
                    "over the kernels of a bench stage; f = the counter's value per byte really moved for the kernel's access width, "
                    "measured in the same run by tools/fetch_calib.hip (profiles/%s_pmc_calibration.json)" % (config, tag, tag)}
pmc = {"_note": "per bench stage: valu_us = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x 2.4 GHz), the time the stage's launches need if they "
                "only issued VALU instructions; lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) of the stage's "
                "largest kernel; instruction counts are properties of kernel + input, not of the run (profiles/%s_pmc_summary.json)" % tag}
# provenance: bench.py copies these into roofline.counters_from (the counters are not re-measured in a bench run)
import hashlib
lib = os.path.join(root, "orb_slam2_aruco_amd", "liborbfe.so")
commit_file = os.path.join(root, "build", "COMMIT")       # written by tools/gpu.sh before the snapshot leaves (no .git on the GPU box)
ident = {"_profile": tag, "_commit": open(commit_file).read().strip() if os.path.exists(commit_file) else None,
         "_library_sha16": hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16] if os.path.exists(lib) else None}
traffic.update(ident, _file="profiles/traffic_%s.json" % config)
pmc.update(ident, _file="profiles/pmc_stage_%s.json" % config)
for st, ks in stage.items():
    ks = [k for k in ks if k in s]
    if not ks:
        continue
    traffic[st] = int(round(sum(kb(k) * per_step(k) for k in ks)))
    big = max(ks, key=lambda k: s[k].get("SQ_INSTS_VALU", 0) * per_step(k))
    g = lambda n: s[big].get(n, 0.0)
    pmc[st] = {"valu_us": sum(s[k].get("SQ_INSTS_VALU", 0.0) * per_step(k) for k in ks) * 4 / 1024 / 2400,
               "lane_utilisation": g("SQ_THREAD_CYCLES_VALU") / max(64 * g("SQ_ACTIVE_INST_VALU"), 1) if g("SQ_ACTIVE_INST_VALU") else None,
               "valu_per_wave": g("SQ_INSTS_VALU") / max(g("SQ_WAVES"), 1), "kernels": ks}
json.dump(traffic, open(os.path.join(root, "profiles", "traffic_%s.json" % config), "w"), indent=1)
json.dump(pmc, open(os.path.join(root, "profiles", "pmc_stage_%s.json" % config), "w"), indent=1)
print({k: v for k, v in traffic.items() if not k.startswith("_")})
print({k: round(v["valu_us"], 1) for k, v in pmc.items() if not k.startswith("_")})
