"""MI355X-native BTV-L1 multiframe super-resolution (hand-written HIP behind a C ABI).

`superresolution_amd.api` mirrors the reference's SuperResolution / BTV_L1 interface,
`superresolution_amd.ops` exposes the individual solver stages on torch CUDA tensors,
`superresolution_amd.synth` makes seeded synthetic sequences.  The compute lives in
libsrx.so (superresolution_amd/csrc); importing this package does not load it, calling
into it does, and a missing library raises instead of falling back to anything else.
"""

__all__ = ["api", "ops", "synth", "spans"]
