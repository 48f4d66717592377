"""Import the reference's hot-path files in THIS container (never on the GPU box).

TEST INFRASTRUCTURE ONLY.  Follows SURVEY.md Appendix A: the reference's package
``__init__`` files star-import timm / torchvision / fvcore users, so the 14 hot-path
files are loaded one by one with importlib into synthetic packages.  Nothing from
the reference is copied into this repository; this module only *executes* files
where they lie under /root/reference and is a no-op (``available() == False``)
everywhere else.
"""
import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get('SEGFAC_REFERENCE', '/root/reference')
_loaded = None


def available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, 'models', 'build_models.py'))


def _pkg(name):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF_ROOT, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _export(dst, src):
    for k, v in vars(src).items():
        if not k.startswith('_'):
            setattr(dst, k, v)


def load():
    """Returns a namespace with SegmentationModel, engine, losses, metrics, utils of the reference."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError(f'reference not present at {REF_ROOT}')
    sys.dont_write_bytecode = True
    saved = {k: sys.modules.get(k) for k in ('models', 'util', 'engine', 'timm')}
    timm = _pkg('timm')
    timm_models = _pkg('timm.models')
    timm_models.register_model = lambda f: f
    timm.models = timm_models
    models = _pkg('models')
    layers, modules = _pkg('models.layers'), _pkg('models.modules')
    backbones, heads = _pkg('models.backbones'), _pkg('models.heads')
    util = _pkg('util')
    for f in ('conv_module', 'drop_path', 'initialize'):
        _export(layers, _load(f'models.layers.{f}', f'models/layers/{f}.py'))
    _export(modules, _load('models.modules.ppm', 'models/modules/ppm.py'))
    for f in ('mit', 'convnext', 'convnextv2', 'mobilenetv2'):
        _export(backbones, _load(f'models.backbones.{f}', f'models/backbones/{f}.py'))
    for f in ('segformer', 'upernet', 'fpn'):
        _export(heads, _load(f'models.heads.{f}', f'models/heads/{f}.py'))
    heads.MaskRCNNHeads = None          # head_dict references it at import (build_models.py:11)
    _load('models.base_model', 'models/base_model.py')
    bm = _load('models.build_models', 'models/build_models.py')
    utils = _load('util.utils', 'util/utils.py')
    util.utils = utils
    losses = _load('util.losses', 'util/losses.py')
    metrics = _load('util.metrics', 'util/metrics.py')
    util.losses, util.metrics = losses, metrics
    engine = _load('engine', 'engine.py')
    _loaded = types.SimpleNamespace(SegmentationModel=bm.SegmentationModel, build_models=bm, engine=engine,
                                    losses=losses, metrics=metrics, utils=utils, backbones=backbones, heads=heads)
    del saved
    return _loaded


def build_reference_model(backbone, head, nc, state_dict, zero_stochastic=True):
    """SegmentationModel(...) with our numpy-seeded weights; DropPath / Dropout2d rates forced to 0
    on the instance when zero_stochastic."""
    import torch
    ref = load()
    model = ref.SegmentationModel(backbone, num_classes=nc, seg_head=head)
    missing, unexpected = model.load_state_dict(state_dict, strict=True)
    assert not missing and not unexpected
    if zero_stochastic:
        for m in model.modules():
            if type(m).__name__ == 'DropPath':
                if hasattr(m, 'p'):
                    m.p = 0.
                if hasattr(m, 'drop_prob'):
                    m.drop_prob = 0.
            if isinstance(m, torch.nn.Dropout2d):
                m.p = 0.
    return model


def load_schedulers():
    """The reference's scheduler package (scheduler/*.py, vendored timm schedulers).  multistep_lr.py:7 imports the base class
    from timm (not installed here): it is given the reference's OWN vendored copy of that class (scheduler/scheduler_main.py,
    which every other file of the package already uses), so the arithmetic executed is the reference's throughout."""
    if not available():
        raise RuntimeError(f'reference not present at {REF_ROOT}')
    sys.dont_write_bytecode = True
    pkg = _pkg('scheduler')
    main = _load('scheduler.scheduler_main', 'scheduler/scheduler_main.py')
    timm = sys.modules.get('timm') or _pkg('timm')
    ts = _pkg('timm.scheduler')
    tss = _pkg('timm.scheduler.scheduler')
    tss.Scheduler = main.Scheduler
    ts.scheduler = tss
    timm.scheduler = ts
    for f in ('cosine_lr', 'tanh_lr', 'step_lr', 'multistep_lr', 'plateau_lr', 'poly_lr'):
        setattr(pkg, f, _load(f'scheduler.{f}', f'scheduler/{f}.py'))
    fac = _load('scheduler.scheduler_factory', 'scheduler/scheduler_factory.py')
    return fac
