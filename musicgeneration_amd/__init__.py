"""musicgeneration_amd -- MI355X-native Music Transformer hot path.

Only what the autoregressive event-sequence path needs: the HIP kernels + C ABI (csrc/, libmgx.so),
their ctypes binding (_lib, ops) and the host-side mirror of the reference's interface
(network.MusicTransformer, criterion, metrics, data, the event codecs, train.py)."""
__version__ = "0.1.0"
