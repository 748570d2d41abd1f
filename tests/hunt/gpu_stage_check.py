"""Stage-by-stage comparison of the HIP engine against the CPU oracle (run on the GPU box)."""
import os, subprocess, sys, time
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from streamkit_amd import engine, synth
from oracle_lib import OracleModel
import oracle_lib

size = sys.argv[1] if len(sys.argv) > 1 else "tiny"
clips = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0").split(",")]
path = "/tmp/synth_%s.bin" % size
if not os.path.exists(path):
    subprocess.check_call([os.path.join(ROOT, "tools", "make_synth_model"), path, "--size", size])
om = OracleModel(path)
gm = engine.Model(path)
ctx = engine.Context(gm, max_batch=max(4, len(clips)))

def cmp(name, a, b):
    a = np.asarray(a); b = np.asarray(b)
    neq = int((a.view(np.uint32) != b.view(np.uint32)).sum()) if a.dtype == np.float32 else int((a != b).sum())
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    print("%-28s bit-mismatches %d / %d   max|d| %.3e   max|ref| %.3e" % (name, neq, a.size, d.max(), np.abs(b).max()), flush=True)
    return neq

pcm = synth.clip(clips[0])
t = time.time(); mel_o, norg_o = om.log_mel(pcm); t_mel_o = time.time() - t
mel_g, norg_g = ctx.log_mel(pcm)
assert norg_o == norg_g and mel_o.shape == mel_g.shape, (norg_o, norg_g, mel_o.shape, mel_g.shape)
cmp("log-mel", mel_g, mel_o)
x0_o = om.conv_stem(mel_o); x0_g = ctx.conv_stem(pcm)
cmp("conv stem (+pos emb)", x0_g, x0_o)
oracle_lib.debug_enable(True); engine.debug_enable(True)
t = time.time(); enc_o, ck_o, cv_o = om.encode(mel_o); t_enc_o = time.time() - t
t = time.time(); enc_g, ck_g, cv_g = ctx.encode(pcm); t_enc_g = time.time() - t
for nm in ("l0.ln1", "l0.q", "l0.k", "l0.v", "l0.rmax", "l0.rinv", "l0.SP", "l0.att32", "l0.att", "l0.x1", "l0.ln2", "l0.h", "l0.x2"):
    a = engine.debug_get(nm); b = oracle_lib.debug_get(nm)
    if a is None or b is None: print("tap", nm, "missing"); continue
    cmp(nm, a, b)
    if nm == "l0.SP":
        Tp = a.size // 64; A = a.reshape(64, Tp); B = b.reshape(64, Tp)
        bad = np.argwhere(A.view(np.uint32) != B.view(np.uint32))
        print("  S mismatches:", int((A[:32].view(np.uint32) != B[:32].view(np.uint32)).sum()), " P mismatches:", int((A[32:].view(np.uint32) != B[32:].view(np.uint32)).sum()))
        print("  first:", [(int(i), int(j), float(A[i, j]), float(B[i, j])) for i, j in bad[:10]])
    if nm == "l0.att":
        d = gm.hp.n_audio_state
        idx = np.nonzero(a.view(np.uint32) != b.view(np.uint32))[0]
        rows = idx // d; cols = idx % d
        print("  att mismatch rows:", sorted(set(rows.tolist()))[:40], " n_rows", len(set(rows.tolist())))
        print("  heads:", sorted(set((cols // 64).tolist())), " sample:", [(int(r), int(c), float(a[i]), float(b[i])) for r, c, i in list(zip(rows, cols, idx))[:8]])
oracle_lib.debug_enable(False); engine.debug_enable(False)
cmp("encoder out (ln_post)", enc_g, enc_o)
cmp("cross K", ck_g, ck_o); cmp("cross V", cv_g, cv_o)
print("oracle encode %.2fs, gpu encode (incl. copies) %.3fs" % (t_enc_o, t_enc_g))
d = om.decoder(ck_o, cv_o)
prompt = [50258, 50259, 50359]
lg_o = d.step(prompt, 0); lg_g = ctx.decode_logits(prompt)
cmp("decoder logits (prompt)", lg_g, lg_o)
toks = prompt + [50364, 1234, 777, 31000]
lg_o = d.step(toks[3:], 3); lg_g = ctx.decode_logits(toks)
cmp("decoder logits (7 tokens)", lg_g, lg_o)
# full pipeline
pcms = [synth.clip(c) for c in clips]
t = time.time(); res_g = ctx.full_batch(pcms); t_g = time.time() - t
print("gpu full_batch %d clips: %.3fs  timing %s" % (len(clips), t_g, ctx.timing()))
ok = True
for c, pc, rg in zip(clips, pcms, res_g):
    t = time.time(); ro = om.full(pc); dt = time.time() - t
    ids_o = [x[0] for x in ro["tokens"]]; ids_g = [x[0] for x in rg["tokens"]]
    same = ids_o == ids_g and [(s["t0"], s["t1"], s["text"]) for s in ro["segments"]] == [(s["t0"], s["t1"], s["text"]) for s in rg["segments"]]
    plog_same = [x[3] for x in ro["tokens"]] == [x[3] for x in rg["tokens"]]
    print("clip %d: oracle %.1fs  tokens %d vs %d  identical=%s  plog identical=%s  windows %d/%d fallback %d/%d  min_margin %.4g/%.4g"
          % (c, dt, len(ids_o), len(ids_g), same, plog_same, ro["n_windows"], rg["n_windows"], ro["fallback_requested"], rg["fallback_requested"], ro["min_margin"], rg["min_margin"]), flush=True)
    if not plog_same:
        dif = [(i, a[0], a[2], b[2], a[3], b[3]) for i, (a, b) in enumerate(zip(ro["tokens"], rg["tokens"])) if a[3] != b[3] or a[2] != b[2]]
        print("  plog diffs (i, id, p_o, p_g, plog_o, plog_g):", len(dif), dif[:6])
    if not same:
        ok = False
        for i, (a, b) in enumerate(zip(ids_o, ids_g)):
            if a != b:
                print("  first diff at", i, a, b); break
print("ALL IDENTICAL" if ok else "MISMATCH")
