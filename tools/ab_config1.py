"""A/B of two builds of the library on the metric workload (configs[1], one chain): the driver's window (5 warm-up + 20 steps of a fresh
chain) and the steady rate, alternating child processes (developer tool).  usage: ab_config1.py <libA.so> <libB.so> [rounds]"""
import os, subprocess, sys, statistics
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import os, sys, time
sys.path.insert(0, %r)
import __graft_entry__ as g
pkg = g.load_package()
model, target = pkg.data.synthetic_femur_target(n_subdiv=6)
setup = pkg.femur_icp_proposal_registration(model, target, fused=2)
w = []
for rep in range(5):
    ctx = pkg.IcpContext(model, target, device=0)
    ch = pkg.SamplingRegistration(ctx, setup, pkg.initial_parameters(model), seed=1024)
    ch.run(5, want_records=False)
    t0 = time.perf_counter(); ch.run(20); w.append(20 / (time.perf_counter() - t0))
    if rep == 4:
        ch.run(300, want_records=False)
        t0 = time.perf_counter(); ch.run(3000, want_records=False); steady = 3000 / (time.perf_counter() - t0)
    ch.close(); ctx.close()
w.sort()
print("RATE", w[2], steady)
''' % root
libs = [os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2])]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
r20, rs = [[], []], [[], []]
for rd in range(rounds):
    for k in (0, 1):
        out = subprocess.run([sys.executable, "-c", child], env=dict(os.environ, ICP_LIBRARY_PATH=libs[k]), capture_output=True, text=True).stdout
        a, b = out.split("RATE")[1].split()[:2]
        r20[k].append(float(a)); rs[k].append(float(b))
for k in (0, 1):
    print("%s: 20-step window median %.0f it/s (%s) | steady %.0f (%s)" % (os.path.basename(libs[k]), statistics.median(r20[k]),
          " ".join("%.0f" % v for v in r20[k]), statistics.median(rs[k]), " ".join("%.0f" % v for v in rs[k])))
print("B / A: window %.4f, steady %.4f" % (statistics.median(r20[1]) / statistics.median(r20[0]), statistics.median(rs[1]) / statistics.median(rs[0])))
