"""HBM traffic per optimiser step and per kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace
only) of tools/prof_step.py: the dispatches between the two marker launches are three whole steps.
usage: pmc_step_summary.py <fetch dir> <write dir> <out.json> [steps=3]
FETCH_SIZE / WRITE_SIZE are KiB per dispatch; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM section): doubled here, WRITE_SIZE as is."""
import csv, glob, json, os, re, sys


def rows_between(d, counter, marker="rng_advance_kernel"):
    f = [d] if os.path.isfile(d) else glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    rows = [r for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    return rows[idx[0] + 1:idx[1]]


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name)


steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
fetch, write = rows_between(sys.argv[1], "FETCH_SIZE"), rows_between(sys.argv[2], "WRITE_SIZE")
per = {}
for rows, key, mul in ((fetch, "fetch", 2.0), (write, "write", 1.0)):
    for r in rows:
        k = short(r["Kernel_Name"])
        e = per.setdefault(k, {"fetch": 0.0, "write": 0.0, "n_fetch": 0, "n_write": 0})
        e[key] += float(r["Counter_Value"]) * 1024.0 * mul
        e["n_" + key] += 1
kern = {}
for k, e in per.items():
    n = max(e["n_fetch"], e["n_write"])
    kern[k] = {"launches_per_step": n / steps, "bytes_per_launch": (e["fetch"] + e["write"]) / max(n, 1),
               "fetch_bytes_per_step": e["fetch"] / steps, "write_bytes_per_step": e["write"] / steps}
tot_f = sum(e["fetch"] for e in per.values()) / steps
tot_w = sum(e["write"] for e in per.values()) / steps
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on tools/prof_step.py",
       "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B), KiB -> bytes", "steps": steps,
       "step_bytes": tot_f + tot_w, "step_fetch_bytes": tot_f, "step_write_bytes": tot_w,
       "kernels": dict(sorted(kern.items(), key=lambda kv: -(kv[1]["fetch_bytes_per_step"] + kv[1]["write_bytes_per_step"])))}
json.dump(res, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "kernels"}))
for k, v in list(res["kernels"].items())[:12]:
    print("%-60s %6.1f launches/step  %9.2f MB/launch  %9.1f MB/step" % (k[:60], v["launches_per_step"], v["bytes_per_launch"] / 1e6,
                                                                       (v["fetch_bytes_per_step"] + v["write_bytes_per_step"]) / 1e6))
