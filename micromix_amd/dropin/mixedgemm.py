"""Put this directory on sys.path in place of the reference's `./mgemm/build/`
(model/qLinearLayer.py:5) and `import mixedgemm` resolves to the MI355X implementation."""
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
from micromix_amd.mixedgemm import *  # noqa: F401,F403,E402
from micromix_amd.mixedgemm import (  # noqa: F401,E402
    activate_quantize_x, downproj_quantize_w, downproj_quantize_w4, rmsnorm_quantize_x,
    batch_decode_i4, batch_decode_f16, init_kv_i4, init_kv_f16, append_kv_i4, append_kv_f16)
