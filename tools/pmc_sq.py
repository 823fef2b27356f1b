"""MFMA / VALU / LDS utilisation table from two rocprofv3 --pmc passes (north_star: "MFMA-utilisation counters").

    python tools/pmc_sq.py PASS_A.db PASS_B.db [kernel-substring ...]   ->  markdown table + JSON (stdout)

pass A: SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
pass B: SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
Derived (MI355X_MICROARCH.md, "rocprofv3 PMC slots" / cycle constants): SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles
summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts SIMD cycles the matrix pipe is busy, summed over the chip's 1024 SIMDs;
GRBM_GUI_ACTIVE = cycles the launch was on the chip.
  kernel_cycles      = SQ_BUSY_CYCLES / 32   (the counter is summed over the chip's 32 shader engines; it matches the un-profiled
                       kernel duration x shader clock, whereas GRBM_GUI_ACTIVE -- summed over 8 XCDs -- also covers the
                       profiler's per-dispatch overhead: wgemm4 50 k vs 71 k cycles)
  mfma_util          = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x kernel_cycles)
  wave_wait_frac     = SQ_WAIT_ANY / SQ_WAVE_CYCLES        (parked on s_waitcnt / barrier)
  wave_issue_stall   = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES   (ready but not issued)
  lds_conflict_frac  = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
"""
import json, sqlite3, sys


def load(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    out = {}
    for name, ctr, n, avg in db.execute(f"select {name_col}, counter_name, count(*), avg(value) from counters_collection group by {name_col}, counter_name"):
        out.setdefault(name, {})[ctr] = (n, avg)
    return out


def rows_from(a, b, subs):
    """Derived per-kernel rows (see the module docstring) from the two passes' tables for kernels whose name contains one of ``subs``."""
    rows = []
    for name in sorted(set(a) | set(b)):
        if not any(s in name for s in subs):
            continue
        ca, cb = a.get(name, {}), b.get(name, {})
        g = lambda d, k: d.get(k, (0, None))[1]
        gui = g(cb, "GRBM_GUI_ACTIVE") or g(ca, "GRBM_GUI_ACTIVE")
        wave = g(ca, "SQ_WAVE_CYCLES")
        kc = (g(ca, "SQ_BUSY_CYCLES") / 32.0) if g(ca, "SQ_BUSY_CYCLES") else (gui / 8.0 if gui else None)
        r = {"kernel": name[:70], "launches": (ca or cb).get("GRBM_GUI_ACTIVE", (0, 0))[0], "kernel_cycles": kc, "gui_cycles_per_xcd": gui / 8.0 if gui else None,
             "mfma_util": (g(cb, "SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * kc)) if kc and g(cb, "SQ_VALU_MFMA_BUSY_CYCLES") is not None else None,
             "lds_busy": (g(cb, "SQ_LDS_IDX_ACTIVE") / (256.0 * kc)) if kc and g(cb, "SQ_LDS_IDX_ACTIVE") is not None else None,
             "insts_mfma": g(cb, "SQ_INSTS_MFMA"), "insts_valu": g(cb, "SQ_INSTS_VALU"), "insts_lds": g(cb, "SQ_INSTS_LDS"),
             "wave_wait_frac": (g(ca, "SQ_WAIT_ANY") / wave) if wave and g(ca, "SQ_WAIT_ANY") is not None else None,
             "wave_issue_stall_frac": (g(ca, "SQ_WAIT_INST_ANY") / wave) if wave and g(ca, "SQ_WAIT_INST_ANY") is not None else None,
             "wave_active_frac": (g(ca, "SQ_ACTIVE_INST_ANY") / wave) if wave and g(ca, "SQ_ACTIVE_INST_ANY") is not None else None,
             "valu_active_frac": (g(ca, "SQ_ACTIVE_INST_VALU") / wave) if wave and g(ca, "SQ_ACTIVE_INST_VALU") is not None else None,
             "lds_conflict_frac": (g(cb, "SQ_LDS_BANK_CONFLICT") / g(cb, "SQ_LDS_IDX_ACTIVE")) if g(cb, "SQ_LDS_IDX_ACTIVE") else None,
             "wait_inst_lds_per_wave_cycle": (g(cb, "SQ_WAIT_INST_LDS") / wave) if wave and g(cb, "SQ_WAIT_INST_LDS") is not None else None}
        rows.append(r)
    return rows


def main():
    a, b = load(sys.argv[1]), load(sys.argv[2])
    subs = sys.argv[3:] or ["wgemm4", "wgemm3", "dense8_kernel", "fd_stage1", "skip_rmsnorm_partials", "moe_gemm"]
    rows = rows_from(a, b, subs)
    f = lambda v, p=3: "-" if v is None else (f"{v:.{p}f}" if isinstance(v, float) else str(v))
    print("| kernel | launches | kernel cycles | MFMA util | LDS busy | wave wait | issue stall | wave active | VALU active | LDS conflict | MFMA / VALU / LDS insts |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print(f"| `{r['kernel']}` | {r['launches']} | {f(r['kernel_cycles'], 0)} | {f(r['mfma_util'])} | {f(r['lds_busy'])} | {f(r['wave_wait_frac'])} | {f(r['wave_issue_stall_frac'])} | {f(r['wave_active_frac'])} | "
              f"{f(r['valu_active_frac'])} | {f(r['lds_conflict_frac'], 4)} | {f(r['insts_mfma'], 0)} / {f(r['insts_valu'], 0)} / {f(r['insts_lds'], 0)} |")
    print()
    print(json.dumps(rows))
    import os
    if os.environ.get("PMC_SQ_JSON"):  # merged store read by bench.py (roofline.mfma_util: a stored profile value, labelled so)
        path = os.environ["PMC_SQ_JSON"]
        store = json.load(open(path)) if os.path.exists(path) else {}
        for r in rows:
            store[r["kernel"]] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if k != "kernel"}
        json.dump(store, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
