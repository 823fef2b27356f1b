"""profiles/pmc_traffic.json from the two PMC passes of tools/pmc_round.sh (FETCH_SIZE and WRITE_SIZE each in its own
rocprofv3 --pmc run over the decode step's eager launches): HBM-side bytes of the W4A16 projections and of the decode attention.
FETCH_SIZE (KB) x 1024 x 2 (the gfx950 correction of MI355X_MICROARCH.md, "HBM"); WRITE_SIZE (KB) x 1024, uncalibrated.
bench.py prints `roofline.traffic` = fetch + write from this file (a stored profile value, labelled so).

    python tools/pmc_traffic.py FETCH.db WRITE.db TAG       # TAG e.g. r06_v1: names the summaries under profiles/
    python tools/pmc_traffic.py --from-json OLD.json TAG    # recompute the summary from a stored per-kernel table
"""
import json, os, sqlite3, sys


def per_kernel(path, counter, subs):
    """{kernel name: (launches, mean counter value)} of the kernels whose name contains one of ``subs``."""
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    out = {}
    for sub in ([subs] if isinstance(subs, str) else subs):
        rows = db.execute(f"select {name_col}, count(*), avg(value) from counters_collection where counter_name = ? and {name_col} like ? "
                          f"group by {name_col}", (counter, f"%{sub}%")).fetchall()
        out.update({r[0]: (r[1], r[2]) for r in rows})
    return out


def weighted(d):
    n = sum(c for c, _ in d.values())
    return (sum(c * v for c, v in d.values()) / n, n) if n else (None, 0)


def per_projection(d):
    """Mean over the step's projection KINDS (one kernel instance each since round 6: q|k|v and o on wss_kernel, gate|up on
    wgemm4_kernel, down on wgemm3_kernel) -- NOT weighted by launch counts (the bench's roofline replays add launches of some
    kinds); the 16-bit lm_head (wgemm16_rows) is not one of the W4A16 projections."""
    kinds = {k: v for k, (c, v) in d.items() if "wgemm16" not in k}
    return (sum(kinds.values()) / len(kinds), len(kinds)) if kinds else (None, 0)


def summarise(gf, gw, af, aw, tag):
    f_kb, n_f = per_projection(gf)
    w_kb, _ = per_projection(gw)
    out = {
        "wgemm_fetch_bytes_per_launch": int(f_kb * 1024 * 2) if f_kb else None,
        "wgemm_write_bytes_per_launch": int(w_kb * 1024) if w_kb else None,
        "algorithmic_bytes_per_launch": 32772096,
        "gfx950_correction": 2.0,
        "per_kernel_KB": {"FETCH_SIZE": {k[:60]: [c, round(v, 1)] for k, (c, v) in gf.items()},
                          "WRITE_SIZE": {k[:60]: [c, round(v, 1)] for k, (c, v) in gw.items()}},
        "launches": f"mean over the {n_f} W4A16 launch kinds of a decoder layer (fused q|k|v partials and o partials: short-stream "
                    "engine; fused gate|up + swiglu: row-group engine; down partials: unit loop; M = 64), eager launches of the step",
        "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python bench.py --no-graph --no-cpu-baseline --no-secondary "
                   "--steps 3 --warmup 1 (WRITE_SIZE in its own pass)",
        "summary": f"profiles/{tag.split('_')[0]}_pmc_FETCH_SIZE_gemm_{tag.split('_')[1]}.txt, "
                   f"profiles/{tag.split('_')[0]}_pmc_WRITE_SIZE_gemm_{tag.split('_')[1]}.txt",
    }
    out["wgemm_bytes_per_launch"] = (out["wgemm_fetch_bytes_per_launch"] or 0) + (out["wgemm_write_bytes_per_launch"] or 0)
    a_f, _ = weighted(af)
    a_w, _ = weighted(aw)
    if a_f:
        out["attention_fetch_bytes_per_launch"] = int(a_f * 1024 * 2)
        out["attention_write_bytes_per_launch"] = int(a_w * 1024) if a_w else None
    return out


def main():
    if sys.argv[1] == "--from-json":
        old = json.load(open(sys.argv[2]))
        tup = lambda d: {k: (c, v) for k, (c, v) in d.items()}
        out = summarise(tup(old["per_kernel_KB"]["FETCH_SIZE"]), tup(old["per_kernel_KB"]["WRITE_SIZE"]), {}, {}, sys.argv[3])
        for k in ("attention_fetch_bytes_per_launch", "attention_write_bytes_per_launch"):
            if k in old:
                out[k] = old[k]
    else:
        fetch_db, write_db, tag = sys.argv[1], sys.argv[2], sys.argv[3]
        gemms = ["wgemm", "wss_kernel"]  # row-group / unit-loop engines and the short-stream engine (round 6: q|k|v and o)
        out = summarise(per_kernel(fetch_db, "FETCH_SIZE", gemms), per_kernel(write_db, "WRITE_SIZE", gemms),
                        per_kernel(fetch_db, "FETCH_SIZE", "fd_stage1"), per_kernel(write_db, "WRITE_SIZE", "fd_stage1"), tag)
        json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
