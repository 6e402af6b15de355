"""Drop-in name for the reference's Python module: `from arbplf import arbplf_ll`."""
from phyly_amd.arbplf import *  # noqa: F401,F403
from phyly_amd.arbplf import (arbplf_ll, arbplf_deriv, arbplf_marginal, arbplf_hess, arbplf_inv_hess,  # noqa: F401
                              arbplf_dwell, arbplf_trans, arbplf_em_update, arbplf_newton_delta,
                              arbplf_newton_update, arbplf_newton_refine)
