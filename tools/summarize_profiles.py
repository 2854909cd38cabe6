"""Turn raw rocprofv3 output (under gpurun_out/) into the small summaries committed under profiles/.

    python tools/summarize_profiles.py stats  <kernel_stats.csv> <out.csv> "<command line that was profiled>"
    python tools/summarize_profiles.py pmc    <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> "<note>"

The commands that produce the inputs (on the GPU box):
    cd /tmp && export TMPDIR=/tmp
    TTS_NO_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_e2e -o e2e -- python $R/tools/e2e_once.py 80
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o f -- python $R/tools/diff_prof.py 2
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o w -- python $R/tools/diff_prof.py 2
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_mfma -o m -- python $R/tools/diff_prof.py 2
    python tools/summarize_profiles.py mfma   <m_counter_collection.csv> <out.json> "<note>"
"""
import csv
import json
import sys
from collections import defaultdict


def short(name):
    return name if len(name) <= 90 else name[:90]


def stats(src, dst, cmd):
    rows = list(csv.DictReader(open(src)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    with open(dst, "w") as f:
        f.write("# %s\n# total kernel time %.1f ms\nkernel,calls,total_ms,avg_us,pct\n" % (cmd, total / 1e6))
        for r in rows[:40]:
            t = float(r["TotalDurationNs"])
            f.write('"%s",%s,%.2f,%.2f,%.2f\n' % (short(r["Name"]), r["Calls"], t / 1e6, float(r["AverageNs"]) / 1e3, 100 * t / total))


def counter_avg(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = acc[r["Kernel_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return {k: (n, s / n) for k, (n, s) in acc.items()}


def pmc(fetch_csv, write_csv, dst, note):
    fe, wr = counter_avg(fetch_csv, "FETCH_SIZE"), counter_avg(write_csv, "WRITE_SIZE")
    out = {"_note": note, "kernels": {}}
    for k, (n, f) in sorted(fe.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        if not k.startswith(("void tts::", "tts::")):
            continue
        w = wr.get(k, (0, 0.0))[1]
        # MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports exactly half the bytes of wide (16 B/lane) coalesced
        # reads, global_load and LDS-DMA alike -> doubled; WRITE_SIZE calibrated x1 (see the note in the output file)
        out["kernels"][short(k)] = {"dispatches": n, "fetch_KiB_raw": round(f, 1), "write_KiB_raw": round(w, 1),
                                    "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
    json.dump(out, open(dst, "w"), indent=1)


def decode(fetch_csv, dst, note, layers=30):
    """FETCH_SIZE pass over the decode step launched kernel by kernel (TTS_NO_GRAPH=1 python tools/ar_decode_only.py N): bytes fetched per launch of
    each decode kernel (x 2: MI355X_MICROARCH.md, HBM) and per step = layers x the five layer kernels + the head (DEC_LOGITS = dec_ln_gemv_kernel<2, ...>)."""
    fe = counter_avg(fetch_csv, "FETCH_SIZE")
    per_launch, step = {}, 0.0
    for k, (n, f) in fe.items():
        if not any(t in k for t in ("dec_ln_gemv_kernel", "dec_gemv_resid_kernel", "attn_decode")):
            continue
        if "attn_decode" not in k and ", true>(" not in k:  # the prompt pass runs the same kernels on the [row][1024] layout (last template flag false): not a decode step
            continue
        mib = 2 * f / 1024.0
        per_launch[short(k)] = {"dispatches": n, "fetch_MiB_per_launch": round(mib, 2)}
        head = "dec_ln_gemv_kernel<2" in k
        step += mib * (1 if head else layers)
    json.dump({"_note": note, "fetch_MiB_per_launch": per_launch, "launches_per_step": {"per_layer": 5, "layers": layers, "head": 1},
               "fetch_bytes_per_step": int(step * 1024 * 1024)}, open(dst, "w"), indent=1)


def mfma(src, dst, note):
    """MFMA utilisation per kernel = SQ_VALU_MFMA_BUSY_CYCLES / (cycles per SIMD x 1024 SIMDs); GRBM_GUI_ACTIVE is summed
    over the 8 XCDs (its per-dispatch value / duration = 8 x the shader clock), so cycles per SIMD = GRBM_GUI_ACTIVE / 8."""
    acc = defaultdict(lambda: defaultdict(float))
    cnt, dur, seen = defaultdict(int), defaultdict(float), set()
    for r in csv.DictReader(open(src)):
        k = r["Kernel_Name"]
        if not k.startswith(("void tts::", "tts::")):
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"]))
            cnt[k] += 1
            dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    out = {"_note": note, "kernels": {}}
    for k in sorted(acc, key=lambda k: -dur[k]):
        n, a = cnt[k], acc[k]
        if a["SQ_VALU_MFMA_BUSY_CYCLES"] <= 0:
            continue
        busy, gui, us = a["SQ_VALU_MFMA_BUSY_CYCLES"] / n, a["GRBM_GUI_ACTIVE"] / n, dur[k] / n / 1e3
        out["kernels"][short(k)] = {"dispatches": n, "avg_us": round(us, 1), "mfma_busy_cycles": int(busy),
                                    "shader_clock_MHz": round(gui / 8 / us), "mfma_util": round(busy / (gui / 8 * 1024), 4)}
    json.dump(out, open(dst, "w"), indent=1)


def lds(src, dst, note):
    """LDS activity per kernel: SQ_LDS_IDX_ACTIVE (LDS-array cycles, summed over the CUs) against the CU-cycles of the dispatch
    (GRBM_GUI_ACTIVE / 8 x 256 CUs), bank-conflict cycles, LDS instructions and the waves' LDS issue stalls."""
    acc = defaultdict(lambda: defaultdict(float))
    cnt, dur, seen = defaultdict(int), defaultdict(float), set()
    for r in csv.DictReader(open(src)):
        k = r["Kernel_Name"]
        if not k.startswith(("void tts::", "tts::")):
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"]))
            cnt[k] += 1
            dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    out = {"_note": note, "kernels": {}}
    for k in sorted(acc, key=lambda k: -dur[k])[:12]:
        n, a = cnt[k], acc[k]
        gui = a["GRBM_GUI_ACTIVE"] / n
        row = {"dispatches": n, "avg_us": round(dur[k] / n / 1e3, 1)}
        for c in sorted(a):
            row[c] = int(a[c] / n)
        if gui > 0 and a.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
            row["lds_active_frac_of_cu_cycles"] = round(a["SQ_LDS_IDX_ACTIVE"] / n / (gui / 8 * 256), 4)
        out["kernels"][short(k)] = row
    json.dump(out, open(dst, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "decode":
        decode(*sys.argv[2:5])
    elif sys.argv[1] == "lds":
        lds(*sys.argv[2:5])
    elif sys.argv[1] == "mfma":
        mfma(*sys.argv[2:5])
    elif sys.argv[1] == "stats":
        stats(*sys.argv[2:5])
    elif sys.argv[1] == "pmc":
        pmc(*sys.argv[2:6])
    else:
        sys.exit(__doc__)
