"""odevio_amd - MI355X-native implementation of the ODE-VIO latent-dynamics hot path.

Host side: Python mirror of the reference's ``DeepVIO`` / ``scripts/config.py`` surface.
Device side: libodevio.so (hand-written HIP for gfx950) behind the C ABI in ``include/odevio.h``.
"""
from .config import default_opt, get_args  # noqa: F401


def __getattr__(name):
    # DeepVIO pulls in the HIP library; keep `import odevio_amd` light for config-only users
    if name == "DeepVIO":
        from .deepvio import DeepVIO
        return DeepVIO
    raise AttributeError(name)
