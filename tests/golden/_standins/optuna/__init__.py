"""Import stand-in: SubGNN/train_config.py imports optuna at module level; the golden harness only calls the
module's pure helpers (read_json, get_hyperparams_optuna) with its own fixed trial."""
import types

samplers = types.ModuleType('optuna.samplers')
samplers.TPESampler = object
integration = types.ModuleType('optuna.integration')
integration.PyTorchLightningPruningCallback = object
import sys as _sys
_sys.modules['optuna.samplers'] = samplers
_sys.modules['optuna.integration'] = integration
