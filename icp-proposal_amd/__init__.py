"""icp-proposal_amd — MI355X-native closest-point-proposal path.

The directory name carries a hyphen (it mirrors the reference repository's name), so import it through
`tests/conftest.py:load_package()` / `__graft_entry__.load_package()`, which register it as `icp_proposal_amd`.
"""
from . import data  # noqa: F401
from . import _native  # noqa: F401
from .api import *  # noqa: F401,F403
from .api import (BatchedStepTicket, expect_contexts, chain_eval_step, chain_step, chain_step_batched, chain_step_prelaunch, initial_parameters, posterior_variability,
                  evaluate_reconstruction_to_ground_truth)  # noqa: F401
from . import sampling  # noqa: F401
from .sampling import (SamplingRegistration, ChainSetup, femur_icp_proposal_registration, femur_random_init_comparison,
                       bfm_fitting_partial, random_initial_parameters, run_chains_batched)  # noqa: F401
from . import sharding  # noqa: F401
from . import loggers  # noqa: F401
