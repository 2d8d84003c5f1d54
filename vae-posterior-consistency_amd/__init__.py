"""MI355X-native (gfx950) VAE posterior-consistency training step.

Drop-in for the hot path of stschia/VAE-posterior-consistency: the model classes keep the reference's
Python API (src/models/VAE.py) and run on hand-written HIP kernels through the C ABI in include/vpc.h.
The directory name is not a Python identifier; import it as `import vpc_amd` (shim at the repo root).
"""
from . import _lib
from ._lib import VpcError, LIB_PATH
from .models import Reg_VAE, vanilla_VAE, Reg_VAE_mask, vanilla_VAE_mask, MAX_EPOCH
from .fused import FusedTrainer
from .notmiwae import REG_notMIWAE_v2, notMIWAE_myversion, NMTrainer
from .harness import (create_missing_uci, create_missing_uci_drop_eddi, model_loader, checkpoint_path, train, eval_vae, result_paths, eval_vae_mnar,
                      mnar_result_path)
from . import notmiwae
from . import eddi
from .eddi import Reg_EDDI, vanilla_EDDI, EDDITrainer
from . import ops
from . import dist as dp
from . import active
from . import wide
from .wide import WideTrainer
from .active import (reward_matrix, R_lindley_chain, chaini_I, chaini_II, active_learning_func, active_result_paths,
                     mc_forward)

__all__ = ["Reg_VAE", "vanilla_VAE", "Reg_VAE_mask", "vanilla_VAE_mask", "FusedTrainer", "REG_notMIWAE_v2",
           "notMIWAE_myversion", "NMTrainer", "notmiwae", "eddi", "Reg_EDDI", "vanilla_EDDI", "EDDITrainer", "eval_vae_mnar", "mnar_result_path", "create_missing_uci", "create_missing_uci_drop_eddi", "model_loader", "checkpoint_path", "train", "eval_vae", "result_paths",
           "VpcError", "ops", "dp", "LIB_PATH", "MAX_EPOCH", "active", "reward_matrix", "R_lindley_chain", "chaini_I",
           "chaini_II"]
