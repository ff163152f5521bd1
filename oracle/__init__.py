"""CPU oracle for the MI355X speaker-embedding + assignment hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker / the timed CPU baseline.  The
product path (``speaker-diarization-toolkit_amd``) never imports this package and
fails loudly when ``libsdk_hip.so`` or a GPU is missing.

PARITY STATUS
-------------
* Plumbing: pinned.  The product's host code is
  checked directly against golden vectors captured by running the reference's
  own Python in the build container (tests/golden/make_golden.py).
* Arithmetic (k1 fbank, k2 ECAPA-TDNN forward, k3 L2-normalise, k4 cosine
  affinity + argmax, k6 spectral clustering): **parity unpinned**.  The
  reference (CLIAI/speaker-diarization-toolkit) contains no implementation of
  this arithmetic and pins no third-party version of one (SURVEY.md §0, §8c:
  SpeechBrain / pyannote are named in prose only -
  speaker_detection.README.md:216-219, backends.yaml:22-31).  The modules here
  are therefore this build's own restatement of the *published* algorithms
  (log-mel filterbank; ECAPA-TDNN, Desplanques et al. 2020, C=1024 layer table
  of SURVEY.md Appendix B; cosine scoring; normalised-Laplacian spectral
  clustering), cross-checked against independent library implementations
  (torch.stft / torch.nn.functional.conv1d / scipy.linalg.eigh / sklearn) in
  tests/test_oracle_*.py.

Numerical contract shared with the HIP kernels ("bf16 layer-boundary model",
DESIGN.md §3): operands of every MFMA GEMM are bf16 (round-to-nearest-even),
accumulation is fp32 (the oracle accumulates in fp64 = the ideal result),
epilogues are fp32, tensors crossing a layer boundary are stored as bf16;
pooling statistics, SE gates, the final FC, L2-normalise and the reported
cosine scores are fp32.
"""
