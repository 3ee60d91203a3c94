"""Per-kernel HBM traffic table from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in separate passes, as the TCC slot budget of
gfx950 requires) joined with the kernel-trace durations of an un-instrumented run of the same command.

    python tools/hbm_table.py <fetch_dir> <write_dir> <trace_dir> <out_csv> [<out_json> <tag> <iterations in the run>]

Corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of
a wide coalesced streaming read (16 B per lane), so the read side is doubled for kernels that read with 16-byte accesses; the
factor is calibrated here on the fused Adam kernel, whose byte count is known (16 B read + 12 B written per parameter, float4
accesses): calibration = 16 * n_params / (FETCH_SIZE * 1024).  Narrower access patterns are uncalibrated — both the raw and the
corrected read figures are written."""
import collections, csv, glob, json, os, sys


def load_pmc(d, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                a = agg[r["Kernel_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    return agg


def load_trace(d):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            a = agg[r["Kernel_Name"]]
            a[0] += 1
            a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    return agg


def main():
    fd, wd, td, out_csv = sys.argv[1:5]
    fetch, write, trace = load_pmc(fd, "FETCH_SIZE"), load_pmc(wd, "WRITE_SIZE"), load_trace(td)
    # calibration on adam_k (params per launch from the arena sizes is not known here: use WRITE_SIZE, exact for 16-B stores:
    # 12 B written per parameter -> reads must be 16/12 of the writes)
    cal = None
    for k in fetch:
        if "adam_k" in k and k in write and fetch[k][1] > 0:
            cal = (write[k][1] / write[k][0]) * 16.0 / 12.0 / (fetch[k][1] / fetch[k][0])
    rows = []
    for k, (n, secs) in trace.items():
        if k not in fetch or k not in write or n == 0:
            continue
        f_kib = fetch[k][1] / fetch[k][0]
        w_kib = write[k][1] / write[k][0]
        t = secs / n
        raw = (f_kib + w_kib) * 1024 / t / 1e9
        cor = (2 * f_kib + w_kib) * 1024 / t / 1e9
        rows.append((secs, k, n, t * 1e6, f_kib * 1024, w_kib * 1024, raw, cor))
    rows.sort(reverse=True)
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "avg_us", "FETCH_SIZE_bytes_per_launch(raw)", "WRITE_SIZE_bytes_per_launch", "GBps_raw",
                    "GBps_fetch_x2(16B-read correction)", "adam_calibration_factor=%s" % (None if cal is None else round(cal, 3))])
        for secs, k, n, us, fb, wb, raw, cor in rows:
            w.writerow([k.replace("(anonymous namespace)::", "")[:160], n, round(us, 2), round(fb), round(wb), round(raw, 1), round(cor, 1)])
    print("adam calibration factor (expected ~2):", cal)
    for secs, k, n, us, fb, wb, raw, cor in rows[:25]:
        print(f"{k.replace('(anonymous namespace)::', '')[:70]:70s} n={n:5d} {us:8.1f} us  fetch {fb/1e6:8.2f} MB write {wb/1e6:8.2f} MB  raw {raw:7.1f}  x2 {cor:7.1f} GB/s")
    if len(sys.argv) > 6:
        out_json, tag, iters = sys.argv[5], sys.argv[6], int(sys.argv[7])
        fam = ("bn_stats_partial", "bn_stats_final", "bn_stats_single", "bn_stats_from_rows", "bn_running_update", "norm_apply_fwd", "norm_bwd_rows",
               "norm_bwd_channels", "norm_bwd_apply", "norm_bwd_table")
        tot_b = tot_n = 0.0
        for secs, k, n, us, fb, wb, raw, cor in rows:
            if any(x in k for x in fam):
                tot_b += (2 * fb + wb) * n
                tot_n += n
        conv_fam = ("pconv_k", "pbww_k", "igemm_f32", "patch_conv", "small_cout_conv", "few_bww_k", "few_cin_fwd_k", "splitk_epilogue",
                    "slab_reduce", "pack_weights_k", "pack_vert_k", "vert_diag_sum_k", "phase_edge_k", "linear_bww_k", "linear_bwd_data_k",
                    "tap_major_to_w_k", "flip_transpose_w", "transpose_in", "pack_weights_h16_k", "h16_scan_k", "h16_scan_many_k", "pack_many_k")
        conv_b = conv_n = 0.0
        for secs, k, n, us, fb, wb, raw, cor in rows:
            if any(x in k for x in conv_fam):
                conv_b += (2 * fb + wb) * n
                conv_n += n
        data = {}
        if os.path.exists(out_json):
            data = json.load(open(out_json))
        # bench.py's HBM roofline object counts one "launch" per C-ABI call of the family (a call issues 2-3 kernels), so the
        # figure it needs is the family's HBM bytes per training iteration; it divides by its own calls-per-iteration count
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "attribute-guided-image-generation-from-layout_amd"))
        from agl import lib as _L
        _lib = _L.load()
        data[tag] = {"abi": _lib.agl_version(), "split_products": _lib.agl_conv2d_split_products(),
                     "collected_at": os.environ.get("AGL_PROFILE_COMMIT", "unknown"),
                     "bytes_per_iteration": round(tot_b / iters), "kernel_launches_per_iteration": round(tot_n / iters),
                     "conv_bytes_per_iteration": round(conv_b / iters), "conv_kernel_launches_per_iteration": round(conv_n / iters),
                     "source": os.path.relpath(out_csv, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                     "note": "FETCH_SIZE x2 (16-byte reads) + WRITE_SIZE, KiB -> bytes, summed over the normalisation-family kernels"}
        json.dump(data, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
