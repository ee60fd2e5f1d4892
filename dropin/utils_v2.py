"""Drop-in shim: `import utils_v2` resolves to the build's metrics module (ref src/utils_v2.py).  See INTEGRATION.md."""
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from bevfusion_multimodal_3d_object_detection_amd.utils_v2 import *  # noqa: F401,F403,E402
from bevfusion_multimodal_3d_object_detection_amd.utils_v2 import (  # noqa: F401,E402
    calculate_ap, compute_center_distance_matrix, compute_metrics, match_predictions_to_gt)
