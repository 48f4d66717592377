"""segmentation_factory_amd -- MI355X-native (gfx950) hot path behind the Segmentation_Factory plugin API.

Layout: ``csrc/`` hand-written HIP kernels + the C ABI of include/segfac.h (libsegfac_hip.so);
``hip.py`` ctypes binding; ``functional.py`` autograd formulas over those kernels; ``backbones.py`` /
``heads.py`` / ``build_models.py`` / ``engine.py`` / ``metrics.py`` / ``utils.py`` / ``optim.py`` / ``inference.py`` mirror
the reference's interfaces (models/build_models.py, engine.py, util/metrics.py, util/utils.py, estimate_model.py).
Importing the package does not load the shared library; the first kernel call does and fails loudly if it
is missing.
"""
from .build_models import SegmentationModel, head_dict, backbone_registry, register_backbone, register_head  # noqa: F401
from .engine import criterion, criterion_lowres, evaluate, train_one_epoch  # noqa: F401
from .metrics import Metrics  # noqa: F401
from .inference import SemSeg  # noqa: F401   (estimate_model.py's SemSeg for tensors)

__version__ = '0.1.0'
