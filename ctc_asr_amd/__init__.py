"""ctc_asr_amd — MI355X-native CTC acoustic-model training path.

Drop-in for the hot path of mdangschat/ctc-asr (``asr/model.py`` and the files around it):
features -> DS1/DS2 stack -> CTC loss / decode, behind hand-written HIP kernels for gfx950 that
are reached through the C ABI declared in ``include/ctcasr.h``.
"""

__version__ = '0.1.0'
