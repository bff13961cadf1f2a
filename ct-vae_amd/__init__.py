"""ctvae-mi355x: MI355X-native (gfx950) VAE training step behind the ct-vae ``BaseVAE`` API.

Sub-modules (imported lazily so that pure-host utilities such as ``filler`` work without a GPU):

* ``native``      – ctypes binding of ``libctvae_hip.so`` (the C-ABI in ``include/ctvae_hip.h``)
* ``kernels``     – ``torch.autograd.Function`` wrappers that launch the HIP kernels
* ``models``      – mirror of the reference's ``models`` package (``vae_models`` registry)
* ``experiment``  – Lightning-free counterpart of ``experiment.py`` / ``run.py``
* ``ddp``         – gradient-bucket all-reduce over RCCL
"""
__version__ = "0.1.0"
