"""The synthetic HEVC header writer lives in the package (hevcbitstream_amd/hevc_synth.py: bench.py and scripts/ build BASELINE
config 3's 4K30 workload with it without reaching into tests/); the tests keep importing it under this name."""
from hevcbitstream_amd.hevc_synth import *          # noqa: F401,F403
from hevcbitstream_amd.hevc_synth import Synth, annexb, stream_4k30   # noqa: F401
