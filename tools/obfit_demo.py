"""obfit / obpred on the Borehole function, printed (demo and timing aid)."""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, ob_oracle as O
import outerbase_amd as ob
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
numb = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rng = np.random.default_rng(42)
x = rng.random((n, 8)); y = O.borehole8d(x)
t0=time.time(); m = ob.obfit(x, y, numb=numb, seed=1, verbose=1); t1=time.time()
xt = rng.random((200, 8)); pred = ob.obpred(m, xt); yt = O.borehole8d(xt)
print("fit s", t1-t0, "rmse/sd", math.sqrt(np.mean((pred["mean"]-yt)**2))/np.std(yt), "hyp", ob.gethyp(m["om"]), "para", ob.getpara(m["logpdf"]))
z=(pred["mean"]-yt)/np.sqrt(pred["var"]); print("rms z", np.sqrt(np.mean(z**2)))
