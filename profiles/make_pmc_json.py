"""profiles/<tag>_pmc_traffic.json from the rocprofv3 summaries (profiles/summarize.py output) -- per-launch HBM bytes that
bench.py quotes in roofline.traffic / op_rooflines[*].pmc_bytes.  FETCH_SIZE is doubled per the MI355X guide's gfx950 correction
for wide streaming reads (an upper bound where part of the reads are 4-byte gathers); WRITE_SIZE as counted.
usage: python profiles/make_pmc_json.py r3   (reads profiles/r3_streams1_summary.txt, r3_ops_summary.txt, r3_ops_n50000_summary.txt)"""
import json, os, re, sys
HERE = os.path.dirname(os.path.abspath(__file__))
tag = sys.argv[1] if len(sys.argv) > 1 else "r3"


def parse(path):
    """-> {"stats": {kernel: avg_us}, "FETCH_SIZE": {kernel: KiB}, "WRITE_SIZE": {...}, "mfma": {kernel: {ctr: v}}}"""
    out = {"stats": {}, "FETCH_SIZE": {}, "WRITE_SIZE": {}, "mfma": {}}
    sec = None
    for line in open(path):
        if line.startswith("== kernel stats"): sec = "stats"; continue
        if line.startswith("== MFMA pipe"): sec = "mfma"; continue
        m = re.match(r"== (FETCH_SIZE|WRITE_SIZE) per launch", line)
        if m: sec = m.group(1); continue
        line = line.rstrip("\n")
        if sec == "stats":
            m = re.match(r"(.{72}) +(\d+) +([\d.]+) +([\d.]+)", line)
            if m: out["stats"][m.group(1).strip()] = float(m.group(3))
        elif sec in ("FETCH_SIZE", "WRITE_SIZE"):
            m = re.match(r"(.{72}) launches +(\d+) +avg +([\d.]+) KiB", line)
            if m: out[sec][m.group(1).strip()] = float(m.group(3))
        elif sec == "mfma":
            m = re.match(r"(.{72}) (.*)", line)
            if m and "=" in m.group(2):
                out["mfma"][m.group(1).strip()] = {kv.split("=")[0]: float(kv.split("=")[1]) for kv in m.group(2).split()}
    return out


def find(d, prefix):
    for k, v in d.items():
        if k.startswith(prefix):
            return k, v
    return None, None


def entry(p, prefixes, how):
    f = w = 0.0
    names, us = [], 0.0
    for pre in prefixes:
        k, v = find(p["FETCH_SIZE"], pre)
        if k: f += v; names.append(k)
        k, v = find(p["WRITE_SIZE"], pre)
        if k: w += v
        k, v = find(p["stats"], pre)
        if k: us += v
    return {"kernel": " + ".join(names), "fetch_kib_raw": f, "write_kib": w, "traffic_bytes": int((2 * f + w) * 1024),
            "avg_us_kernel_trace": us, "how": how}


res = {}
s1 = os.path.join(HERE, "%s_streams1_summary.txt" % tag)
if os.path.exists(s1):
    p = parse(s1)
    how = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE in "
           "separate passes of `python3 bench.py --steps 10 --warmup 2 --legs headline --streams 1` (profiles/run_profile.sh %s_streams1); "
           "per-launch averages; FETCH_SIZE doubled (gfx950 correction for wide reads: an upper bound), WRITE_SIZE as counted" % tag)
    e = entry(p, ["crop_packed_split_kernel"], how)
    k, m = find(p["mfma"], "crop_packed_split_kernel")
    if m:
        us = e["avg_us_kernel_trace"] or 1.0
        e["mfma"] = dict(m, issued_bf16_tflops=m.get("SQ_INSTS_MFMA", 0) * 32768 / (us * 1e-6) / 1e12,
                         frac_of_2500=m.get("SQ_INSTS_MFMA", 0) * 32768 / (us * 1e-6) / 2.5e15,
                         mfma_busy_share_of_simd_cycles=(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) /
                                                         max(m.get("GRBM_GUI_ACTIVE", 1) / 8.0 * 1024, 1)))
    res["crop_unique_mlp_pool"] = e
for name, fn, spec in (("ops", "%s_ops_summary.txt", {"group_points_sa2": ["group_points_lds_kernel"], "group_points_sa1": ["group_points_vec_small_kernel", "group_points_vec_kernel"],
                                                       "ball_query_sa1_cold": ["grid_build_kernel", "ball_grid_kernel<false>"],
                                                       "cylinder_query_cold": ["grid_build_kernel", "ball_grid_kernel<true>"],
                                                       # round 4: a query on a cloud the previous query binned reuses its grid
                                                       "ball_query_sa1": ["ball_grid_kernel<false>"],
                                                       "cylinder_query": ["ball_grid_kernel<true>"]}),
                       ("ops_n50000", "%s_ops_n50000_summary.txt", {"group_points_sa1_n50000": ["group_points_vec_small_kernel", "group_points_vec_kernel"],
                                                                    "ball_query_sa1_n50000": ["grid_build_kernel", "ball_grid_kernel<false>"],
                                                                    "furthest_point_sampling_n50000": ["fps_cell_kernel"]})):
    path = os.path.join(HERE, fn % tag)
    if not os.path.exists(path):
        continue
    p = parse(path)
    for key, pres in spec.items():
        res[key] = entry(p, pres, "same counters over the drop-in operator calls (profiles/run_profile_ops.sh %s_ops -> profiles/%s)" % (tag, fn % tag))
json.dump(res, open(os.path.join(HERE, "%s_pmc_traffic.json" % tag), "w"), indent=1)
print(json.dumps(res, indent=1)[:1500])
