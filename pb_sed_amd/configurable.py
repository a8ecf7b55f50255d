"""Config -> object plumbing of the model API: ``get_config`` / ``finalize_dogmatic_config`` / ``from_config`` /
``from_storage_dir``, i.e. the part of padertorch's ``Configurable`` / ``Model`` the reference's experiment scripts
lean on (reference pb_sed/models/weak_label/crnn.py:304-340, pb_sed/models/strong_label/crnn.py:155-198,
pb_sed/experiments/weak_label_crnn/inference.py:407-413, pb_sed/experiments/weak_label_crnn/training.py:186-281).

Rules restated from padertorch (absent here, parity unpinned): a config is a nested dict; a dict with a ``'factory'`` key
describes a call ``factory(**other_keys)``; missing keys are filled from the factory's signature defaults; a factory's
``finalize_dogmatic_config(config)`` classmethod may fill in or derive further entries, but whatever the USER put into
the config wins over what the class derives ("dogmatic"); sub-configs are completed and instantiated depth first.
"""
import importlib
import inspect
import json
import os

import torch


def _import_path(obj):
    return obj if isinstance(obj, str) else f'{obj.__module__}.{obj.__qualname__}'


def _reference_factories():
    """Import paths found in config.json files written by the reference's trainer -> the build's classes."""
    from . import modules
    from .models import strong_label, weak_label
    table = {
        'pb_sed.models.weak_label.crnn.CRNN': weak_label.CRNN,
        'pb_sed.models.weak_label.CRNN': weak_label.CRNN,
        'pb_sed.models.strong_label.crnn.CRNN': strong_label.CRNN,
        'pb_sed.models.strong_label.CRNN': strong_label.CRNN,
        'padertorch.contrib.je.modules.features.NormalizedLogMelExtractor': modules.NormalizedLogMelExtractor,
        'padertorch.contrib.je.modules.hybrid.CNN': modules.CNN,
        'padertorch.contrib.je.modules.conv.CNN2d': modules.CNN2d,
        'padertorch.contrib.je.modules.conv.CNN1d': modules.CNN1d,
        'padertorch.contrib.je.modules.rnn.GRU': modules.GRU,
        'paderbox.transform.module_fbank.MelWarping': modules.MelWarping,
        'paderbox.utils.random_utils.LogTruncatedNormal': modules.LogTruncatedNormal,
        'paderbox.utils.random_utils.TruncatedExponential': modules.TruncatedExponential,
        'paderbox.utils.random_utils.Uniform': modules.Uniform,
        'torch.nn.modules.rnn.GRU': torch.nn.GRU,
        'torch.nn.GRU': torch.nn.GRU,
    }
    return table


def resolve_factory(factory):
    """A class / callable, one of the reference's import paths, or any importable dotted path."""
    if not isinstance(factory, str):
        return factory
    table = _reference_factories()
    if factory in table:
        return table[factory]
    module, _, name = factory.rpartition('.')
    return getattr(importlib.import_module(module), name)


def _signature_defaults(factory):
    try:
        sig = inspect.signature(factory)
    except (TypeError, ValueError):
        return {}
    return {n: p.default for n, p in sig.parameters.items()
            if p.default is not inspect.Parameter.empty and p.kind in (p.POSITIONAL_OR_KEYWORD, p.KEYWORD_ONLY)}


class DogmaticConfig:
    """The object a ``finalize_dogmatic_config(config)`` classmethod receives: dict-like, nested, auto-vivifying.
    ``user`` holds what the caller asked for at this level; assignments never override it."""

    def __init__(self, user=None, factory=None):
        self.user = dict(user or {})
        self.data = {}
        for k, v in self.user.items():
            if k != 'factory' and not isinstance(v, dict):
                self.data[k] = v
        f = self.user.get('factory', factory)
        if f is not None:
            self._adopt_factory(f)

    # ---- factory handling
    def _adopt_factory(self, factory):
        factory = resolve_factory(factory)
        self.data['factory'] = factory
        for k, v in _signature_defaults(factory).items():
            if k not in self.data and not isinstance(self.user.get(k), dict):
                self.data[k] = v
        for k, v in self.user.items():                      # user sub-configs become nested configs
            if isinstance(v, dict) and k not in self.data:
                self.data[k] = DogmaticConfig(v)
        fin = getattr(factory, 'finalize_dogmatic_config', None)
        if fin is not None:
            fin(self)

    # ---- dict protocol used by finalize_dogmatic_config implementations
    def __getitem__(self, key):
        if key not in self.data:
            self.data[key] = DogmaticConfig(self.user.get(key) if isinstance(self.user.get(key), dict) else None)
        return self.data[key]

    def __setitem__(self, key, value):
        if key == 'factory':
            if 'factory' not in self.user and self.data.get('factory') is None:
                self._adopt_factory(value)
            return
        user = self.user.get(key)
        if key in self.user and not isinstance(user, dict):
            return                                          # dogma: the user's plain value (possibly None) stays
        if isinstance(value, DogmaticConfig):
            value = value.to_dict()
        if isinstance(value, dict):
            cur = self.data.get(key)
            if not isinstance(cur, DogmaticConfig):
                cur = self.data[key] = DogmaticConfig(user if isinstance(user, dict) else None)
            if 'factory' in value:
                cur['factory'] = value['factory']           # only if neither the user nor an earlier rule chose one
            for k, v in value.items():
                if k != 'factory':
                    cur[k] = v                              # recursion; the user's entries win at every level
            return
        self.data[key] = value

    def __contains__(self, key):
        return key in self.data

    def get(self, key, default=None):
        return self.data[key] if key in self.data else default

    def keys(self):
        return self.data.keys()

    def update(self, other=None, **kw):
        for k, v in dict(other or {}, **kw).items():
            self[k] = v

    def to_dict(self):
        out = {}
        for k, v in self.data.items():
            out[k] = v.to_dict() if isinstance(v, DogmaticConfig) else v
        return out


def _jsonable(cfg):
    if isinstance(cfg, dict):
        return {k: (_import_path(v) if k == 'factory' else _jsonable(v)) for k, v in cfg.items()}
    if isinstance(cfg, (list, tuple)):
        return [_jsonable(v) for v in cfg]
    if isinstance(cfg, torch.Tensor):
        return cfg.tolist()
    if hasattr(cfg, 'item') and getattr(cfg, 'shape', None) == ():
        return cfg.item()
    return cfg


def instantiate(cfg):
    """Depth-first ``factory(**kwargs)`` of a completed config (dicts without a factory stay dicts)."""
    if isinstance(cfg, dict):
        kw = {k: instantiate(v) for k, v in cfg.items() if k != 'factory'}
        if 'factory' in cfg:
            return resolve_factory(cfg['factory'])(**kw)
        return kw
    if isinstance(cfg, list):
        return [instantiate(v) for v in cfg]
    return cfg


class Configurable:
    """Mix-in: ``cls.get_config(updates)`` -> completed dict, ``cls.from_config(config)`` -> instance."""

    @classmethod
    def get_config(cls, updates=None):
        updates = dict(updates or {})
        updates.setdefault('factory', cls)
        return DogmaticConfig(updates).to_dict()

    @classmethod
    def from_config(cls, config):
        config = dict(config)
        config.setdefault('factory', cls)
        return instantiate(config)

    @classmethod
    def from_storage_dir(cls, storage_dir, config_name='config.json', checkpoint_name='ckpt_best_loss.pth',
                         in_config_path='trainer.model', in_checkpoint_path='model', map_location='cpu'):
        """padertorch ``Model.from_storage_dir``: ``<storage_dir>/<config_name>`` holds the experiment config (the
        model's sub-config under ``in_config_path``), ``<storage_dir>/checkpoints/<checkpoint_name>`` the trainer
        checkpoint whose ``in_checkpoint_path`` entry is the model's state_dict."""
        with open(os.path.join(storage_dir, config_name)) as fh:
            config = json.load(fh)
        for part in [p for p in in_config_path.split('.') if p]:
            config = config[part]
        model = cls.from_config(cls.get_config(config))
        ckpt = torch.load(os.path.join(storage_dir, 'checkpoints', checkpoint_name), map_location=map_location,
                          weights_only=False)
        for part in [p for p in in_checkpoint_path.split('.') if p]:
            ckpt = ckpt[part]
        model.load_state_dict(ckpt)
        return model

    def dump_config(self, path, config, in_config_path='trainer.model'):
        """Write ``config`` (the completed model config) the way the reference's trainer stores it."""
        nested = _jsonable(config)
        for part in reversed([p for p in in_config_path.split('.') if p]):
            nested = {part: nested}
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, 'w') as fh:
            json.dump(nested, fh, indent=1)
