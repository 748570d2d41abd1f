import os, sys, numpy as np
os.environ.setdefault("OMP_NUM_THREADS", "16")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import kokoro_lib
d = kokoro_lib.synth_kokoro_dir("micro")
tts = kokoro_lib.Tts(d); orc = kokoro_lib.OracleTts(d); tts.taps(True)
text = "A considerably longer sentence, with commas; semicolons: and colons - so that the token count moves the style row and the frame count grows well past a few hundred frames."
y, _ = tts.generate(text, 7, 1.25); r = orc.synth(text, 7, 1.25)
f0g = tts.tap(1); f0c = r["f0"]
print("f0 straddles 10:", np.nonzero((f0g > 10) != (f0c > 10))[0], "max |df0|", np.abs(f0g - f0c).max())
pg = tts.tap(4).reshape(-1, 22); pc = r["post"]
err = np.sqrt(((pg - pc) ** 2).mean(1)); rms = np.sqrt((pc ** 2).mean())
bad = np.nonzero(err > 1e-3 * rms)[0]
print("rows with err > 1e-3 rms:", bad.size, "of", err.size, "first", bad[:10], "last", bad[-10:])
if bad.size:
    m = bad // 60
    print("f0 index of bad rows:", np.unique(m)[:40])
    print("f0 there (gpu):", f0g[np.unique(m)[:10]], "cpu:", f0c[np.unique(m)[:10]])
print("wave err by second:", [float("%.2g" % np.sqrt(((y[i:i+24000]-r["y"][i:i+24000])**2).mean())) for i in range(0, y.size, 24000)])
hg = tts.tap(8).reshape(-1, 22); hc = r["har"]
dm = np.abs(hg[:, :11] - hc[:, :11]); dp = np.abs(hg[:, 11:] - hc[:, 11:])
print("har: max |dmag|", dm.max(), "rows/bins with |dphase| > 1:", np.argwhere(dp > 1.0)[:20].tolist(), "count", int((dp > 1.0).sum()), "median dphase", np.median(dp), "99.9%", np.quantile(dp, 0.999))
w = np.argwhere(dp > 1e-3)
print("rows with dphase > 1e-3:", len(w), w[:20].tolist())
for (a, b) in w[:8]:
    print(a, b, "gpu mag/phase", hg[a, b], hg[a, 11 + b], "cpu", hc[a, b], hc[a, 11 + b])
