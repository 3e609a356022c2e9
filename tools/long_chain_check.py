"""Developer check: a long femur-50 chain on the GPU against the oracle (tree back end): identical decisions, states within 1e-5."""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import __graft_entry__ as graft
pkg = graft.load_package()
from oracle import oracle
from test_gpu_chain import oracle_chain_config
n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
model, target = pkg.data.load_femur_model_and_target(50)
om, ot = oracle.OracleModel.from_model(model), oracle.OracleMesh(target.points, target.cells)
setup = pkg.femur_icp_proposal_registration(model, target, fused=2)
theta0 = pkg.initial_parameters(model)
oracle.set_search_backend(oracle.SEARCH_TREES)
t = time.time()
acc_o, comp_o, logp_o, states_o = oracle.run_chain(om, ot, oracle_chain_config(oracle, setup), theta0, 1024, n_steps)
print("oracle: %.1f s" % (time.time() - t))
ctx = pkg.IcpContext(model, target, device=0)
chain = pkg.SamplingRegistration(ctx, setup, theta0, 1024)
rec = chain.run(n_steps)
same = np.array_equal(rec[:, 1].astype(np.uint8), acc_o) and np.array_equal(rec[:, 2].astype(np.int32), comp_o)
first = int(np.argmax(rec[:, 1].astype(np.uint8) != acc_o)) if not same else -1
print("decisions identical:", same, "first difference at step", first, "| accepted", int(acc_o.sum()), "of", n_steps)
scale = np.abs(states_o[:, 10:]).max()
upto = n_steps if same else first
print("max state deviation (relative) up to there: %.2e" % (np.abs(rec[:upto, 14:] - states_o[:upto, 10:]).max() / scale))
