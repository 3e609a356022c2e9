"""A/B of two builds of the library on the configs[3] chain (developer tool): alternating child processes, one library each
(ICP_LIBRARY_PATH), the same chain, medians.  usage: ab_config3.py <libA.so>[,K=V…] <libB.so>[,K=V…] [sampler] [steps] [rounds]   (K=V: environment of that side's children)"""
import os, subprocess, sys, statistics
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import os, sys, time
sys.path.insert(0, %r)
import __graft_entry__ as g
pkg = g.load_package()
model = pkg.data.synthetic_face_model()
target = pkg.data.synthetic_partial_target(model, seed=7)
ctx = pkg.IcpContext(model, target, device=0)
setup = pkg.bfm_fitting_partial(model, target, evaluator="hausdorff")
setup.sampler = sys.argv[1]
ch = pkg.SamplingRegistration(ctx, setup, pkg.initial_parameters(model), seed=5)
ch.run(100, want_records=False)
n = int(sys.argv[2])
t0 = time.perf_counter(); ch.run(n, want_records=False); dt = time.perf_counter() - t0
print("RATE", n / dt)
''' % root
sides = [sys.argv[1].split(","), sys.argv[2].split(",")]
libs = [os.path.abspath(sd[0]) for sd in sides]
extra = [dict(kv.split("=", 1) for kv in sd[1:]) for sd in sides]
sampler = sys.argv[3] if len(sys.argv) > 3 else "eigen"
steps = sys.argv[4] if len(sys.argv) > 4 else "1500"
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 4
rates = [[], []]
for rd in range(rounds):
    for k in (0, 1):
        env = dict(os.environ, ICP_LIBRARY_PATH=libs[k], **extra[k])
        out = subprocess.run([sys.executable, "-c", child, sampler, steps], env=env, capture_output=True, text=True).stdout
        rates[k].append(float(out.split("RATE")[1].split()[0]))
for k in (0, 1):
    print("%s %s: median %.1f it/s  (%s)" % (os.path.basename(libs[k]), extra[k], statistics.median(rates[k]), " ".join("%.0f" % v for v in rates[k])))
print("B / A = %.4f" % (statistics.median(rates[1]) / statistics.median(rates[0])))
