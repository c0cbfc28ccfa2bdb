"""ssr-speech_amd — MI355X (gfx950) native implementation of SSR-Speech's inference hot path:
the causal codec-token transformer decode loop (`models/ssr.py::SSR_Speech.inference`) and the
watermarked-Encodec SEANet/LSTM/RVQ codec (`audiocraft/.../wmencodec.py`), as hand-written HIP
kernels behind a C-ABI (`include/ssrhip.h`, built to `ssr-speech_amd/csrc/libssrhip.so`).

Importable as ``ssr_speech_amd`` (see the alias module at the repo root). The HIP library is loaded
lazily by `ssr_speech_amd._lib`; every compute entry point raises if it is missing — there is no
CPU fallback in this package (the CPU restatement lives in `oracle/` and is test-only).
"""
__version__ = "0.1.0"
