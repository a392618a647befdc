"""Make the reference runner's own import lines resolve to the drop-in modules, without editing the reference:

    import neuraludf_amd.dropin; neuraludf_amd.dropin.install()
    import exp_runner_blending          # its `from models.fields import ...`, `from loss.loss import ColorLoss` (:15-21)

`install()` registers `models`, `models.fields`, `models.udf_renderer_blending`, `models.embedder`,
`models.patch_projector`, `loss`, `loss.loss`, `loss.patch_metric` in `sys.modules` as aliases of the packages under
`neuraludf_amd`.  Everything else of the reference (`dataset`, `extract_mesh`, ...) keeps importing from the reference
tree.  `uninstall()` removes the aliases again."""
from __future__ import annotations

import importlib
import sys

_ALIASES = {
    "models": "neuraludf_amd.models",
    "models.fields": "neuraludf_amd.models.fields",
    "models.udf_renderer_blending": "neuraludf_amd.models.udf_renderer_blending",
    "models.embedder": "neuraludf_amd.models.embedder",
    "models.patch_projector": "neuraludf_amd.models.patch_projector",
    "loss": "neuraludf_amd.loss",
    "loss.loss": "neuraludf_amd.loss.loss",
    "loss.patch_metric": "neuraludf_amd.loss.patch_metric",
}
_installed = {}


def install():
    for alias, target in _ALIASES.items():
        if alias in sys.modules and alias not in _installed and not sys.modules[alias].__name__.startswith("neuraludf_amd"):
            raise RuntimeError(f"{alias!r} is already imported from somewhere else; call install() before the runner")
        mod = importlib.import_module(target)
        sys.modules[alias] = mod
        _installed[alias] = mod
    return sorted(_installed)


def uninstall():
    for alias, mod in list(_installed.items()):
        if sys.modules.get(alias) is mod:
            del sys.modules[alias]
        del _installed[alias]
