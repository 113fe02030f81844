"""mfas_amd — MI355X-native inner candidate-training engine for MFAS (jperezrua/mfas).

Only the hot path lives here: the searchable fusion network, its trainer and the population driver,
executed by hand-written gfx950 HIP kernels behind a C ABI (include/mfas_hip.h).
"""
from .engine import (FeatureLoader, FeatureTable, Hyper, Population, best_dev_accuracy, best_dev_f1,  # noqa: F401
                     flat_layout)
from .ntu_searchable import (Searchable_Skeleton_Image_Net, get_central_states,  # noqa: F401
                             get_possible_layer_configurations, set_central_states, train_sampled_models)
from . import avmnist_searchable, mmimdb_searchable  # noqa: F401
from .scheduler import FixedScheduler, LRCosineAnnealingScheduler  # noqa: F401
from .train_ntu import test_ntu_track_acc, train_ntu_track_acc  # noqa: F401

__version__ = "0.1.0"
