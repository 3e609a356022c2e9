"""A/B of library builds on the on-device many-chain loop (developer tool): alternating child processes of tools/r3_device_loop.py,
one library each (ICP_LIBRARY_PATH), medians.  usage: ab_device_loop.py <chains> <steps> <rounds> <lib>[,K=V…] <lib>[,K=V…] …"""
import os, subprocess, sys, statistics
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
chains, steps, rounds = sys.argv[1], sys.argv[2], int(sys.argv[3])
sides = [a.split(",") for a in sys.argv[4:]]
rates = [[] for _ in sides]
for rd in range(rounds):
    for k, sd in enumerate(sides):
        env = dict(os.environ, ICP_LIBRARY_PATH=os.path.abspath(sd[0]), **dict(kv.split("=", 1) for kv in sd[1:]))
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "r3_device_loop.py"), chains, steps, "eigen", "/tmp/ab_dl.npy"],
                             env=env, capture_output=True, text=True)
        try:
            rates[k].append(float(out.stdout.split("(without records ")[1].split(")")[0]))
        except Exception:
            print("FAILED", sd, out.stdout[-300:], out.stderr[-600:]); rates[k].append(float("nan"))
for k, sd in enumerate(sides):
    print("%s: median %.0f it/s  (%s)" % (" ".join(sd), statistics.median(rates[k]), " ".join("%.0f" % v for v in rates[k])))
