"""elektronn2_amd -- MI355X-native hot path behind ELEKTRONN2's neuromancer
Conv / Pool / UpConv node API.  HIP-only: importing
``elektronn2_amd.backend`` without ``libe2hip.so`` raises."""
__version__ = "0.1.0"


def set_plan_options(**kw):
    """host-side switches of the launch plans constructed from now on (neuromancer/options.py)"""
    from .neuromancer.options import set_plan_options as _set
    _set(**kw)


def set_mfma_dtype(dtype):
    """'f32' (default) | 'bf16' -- see neuromancer.plan.set_mfma_dtype"""
    from .neuromancer.plan import set_mfma_dtype as _set
    _set(dtype)
