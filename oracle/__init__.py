"""CPU oracle for the stylish-tts acoustic hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch (fp32, CPU) restatement of the
reference's algorithm for the path SURVEY.md section 8 names.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it, and only as the
checker / reported CPU baseline -- never as the thing measured or shipped.  The product path
(`stylish_tts_amd`) must never import from here and fails loudly without its HIP library.

Parity pinning: the reference has no tests or golden vectors of its own.
The oracle is pinned against outputs of the reference itself, generated in the build container
by `tools/gen_golden.py` (imports /root/reference behind stub modules) and committed under
`tests/golden/`.  Exceptions, stated where they occur: the mel front end and MelScale depend on
torchaudio, which is absent and un-pinned in the reference ("parity unpinned" at that boundary;
cross-checked against torch.stft and the closed-form HTK filter bank instead).

Every function cites the reference file:line it follows (paths relative to
/root/reference/src/stylish_tts/).
"""
