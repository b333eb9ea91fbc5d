"""carta1_amd -- MI355X-native ATRAC1 encode/decode hot path (drop-in for aynik/carta1's
codec/transforms, codec/analysis and codec/coding behind the encode()/decode() closures).

Python host side (the JavaScript host lives in carta1_amd/js).  Everything computes on the GPU through
libcarta1_hip.so; nothing here falls back to the CPU.
"""
from .capi import Carta1Error, FRAME, UNIT_BYTES, SIGNAL_WHITE, SIGNAL_PINK_BURSTS, SIGNAL_MIXED, SIGNAL_PARTIALS  # noqa: F401
from .codec import (Context, EncoderOptions, encode_multi, decode_multi, encode_pcm, decode_units, encode_aea_pcm, decode_aea_pcm,  # noqa: F401
                    EncoderStream, DecoderStream, aea_header, parse_aea_header, pinned_empty)
from .shard import shard_plan, encode_sharded, decode_sharded  # noqa: F401,E402
