"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/prof_decoder_fwd.py into a JSON file
(usage: pmc_summary.py <fetch dir> <write dir> <out.json>; tools/profile_round.sh copies it to profiles/rNN_pmc.json).
FETCH_SIZE / WRITE_SIZE are in KiB... per dispatch; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced
reads (MI355X_MICROARCH.md, HBM section) and is doubled here; WRITE_SIZE is taken as is."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def between_markers(d, counter, marker="rng_advance_kernel"):
    """Sum of `counter` over the dispatches between the first two marker launches (file order = dispatch order)."""
    f = [d] if os.path.isfile(d) else glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    rows = [r for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    return sum(float(r["Counter_Value"]) for r in rows[idx[0] + 1:idx[1]])
def load(d, counter):
    f = [d] if os.path.isfile(d) else glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    rows = list(csv.DictReader(open(f[0])))
    out = {}
    for r in rows:
        if r["Counter_Name"] != counter:
            continue
        out.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return out
fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
Tt, NSEQ = 40, 3
DEC = ("gru_step_kernel", "gru_step_small_kernel", "skinny_plain_kernel", "attn_scores_kernel", "attn_ctx_kernel")
def total(d, pred):
    return sum(sum(v) for k, v in d.items() if pred(k))
def kib(x):
    return x * 1024.0
# decoder step = everything the seq op launches inside its loop (isolated gru-cell launches are the LAST 200 gru_step dispatches)
g_f = [v for k, v in fetch.items() if "gru_step_small_kernel" in k][0]      # decoder-shape cells (M=64, one direction)
g_w = [v for k, v in write.items() if "gru_step_small_kernel" in k][0]
cell_fetch = sum(g_f[-200:]) / 200; cell_write = sum(g_w[-200:]) / 200
# decoder step = every dispatch of the three bracketed sequence calls (4 kernels per step + the per-batch products), / steps
dec_fetch = between_markers(sys.argv[1], "FETCH_SIZE")
dec_write = between_markers(sys.argv[2], "WRITE_SIZE")
b_f = [v for k, v in fetch.items() if "gru_bwd_step_kernel" in k][0]
b_w = [v for k, v in write.items() if "gru_bwd_step_kernel" in k][0]
bwd_fetch = sum(b_f[-200:]) / 200; bwd_write = sum(b_w[-200:]) / 200
def per_launch(name):
    f = [v for k, v in fetch.items() if name in k]
    w = [v for k, v in write.items() if name in k]
    if not f or not w:
        return None
    return kib(2 * sum(f[0]) / len(f[0]) + sum(w[0]) / len(w[0]))
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/prof_decoder_fwd.py",
       "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B), KiB -> bytes",
       "gru_step_kernel_bytes_per_launch": kib(2 * cell_fetch + cell_write),
       "gru_step_kernel_fetch_bytes": kib(2 * cell_fetch), "gru_step_kernel_write_bytes": kib(cell_write),
       "gru_bwd_step_kernel_bytes_per_launch": kib(2 * bwd_fetch + bwd_write),
       "gru_bwd_step_kernel_fetch_bytes": kib(2 * bwd_fetch), "gru_bwd_step_kernel_write_bytes": kib(bwd_write),
       "decoder_step_bytes": kib(2 * dec_fetch + dec_write) / (NSEQ * Tt),
       # round 3: the persistent recurrences (one launch = all steps); keys and weights stay on chip, so the measured traffic
       # is far below the streaming model's algorithmic bytes (Tt x 35.67 MB for the decoder)
       "dec_fwd_persistent_kernel_bytes_per_launch": per_launch("dec_fwd_persistent_kernel"),
       "enc_fwd_persistent_kernel_bytes_per_launch": per_launch("enc_fwd_persistent_kernel"),
       "dec_bwd_persistent_kernel_bytes_per_launch": per_launch("dec_bwd_persistent_kernel"),
       "enc_bwd_persistent_kernel_bytes_per_launch": per_launch("enc_bwd_persistent_kernel")}
out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", "pmc.json")
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
