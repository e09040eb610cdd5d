"""magma_amd: MI355X-native (gfx950, FP64) implementation of MAGMA's multi-task-GP EM hot path.

The public surface mirrors the reference's R functions (MAGMA.R / Computing_functions.R);
everything numeric runs in libmagma_hip.so (hand-written HIP kernels behind a C ABI).
"""
from .api import (kernel, kernel_mu, mat_dist, deriv_hp1, deriv_hp2, dmvnorm, update_inv, update_mean, kern_to_cov, kern_to_inv, logL_GP, gr_GP,
                  logL_GP_mod, gr_GP_mod, logL_GP_mod_common_hp, gr_GP_mod_common_hp,
                  logL_monitoring, e_step, m_step, training, train_new_gp, posterior_mu, pred_gp,
                  full_algo, posterior_mu_resident, pred_gp_batched)
from .engine import Dataset, HipBackend
from .native import NativeEngine
from .lbfgsb import minimize_batch
from .io import read_csv2, split_by

__all__ = ["kernel", "kernel_mu", "mat_dist", "deriv_hp1", "deriv_hp2", "dmvnorm", "update_inv",
           "update_mean", "kern_to_cov", "kern_to_inv", "logL_GP", "gr_GP",
           "logL_GP_mod", "gr_GP_mod", "logL_GP_mod_common_hp", "gr_GP_mod_common_hp",
           "logL_monitoring", "e_step", "m_step", "training", "train_new_gp", "posterior_mu",
           "pred_gp", "full_algo", "posterior_mu_resident", "pred_gp_batched", "Dataset", "HipBackend", "NativeEngine", "minimize_batch",
           "read_csv2", "split_by"]
