"""Scaffolding stub (test infrastructure, never shipped to the GPU box): a minimal
yacs/fvcore-style CfgNode sufficient for importing /root/reference/muvo/config.py."""
import copy
import yaml


class CfgNode(dict):
    def __init__(self, init_dict=None, key_list=None, new_allowed=False):
        super().__init__()
        self.__dict__['_frozen'] = False
        self.__dict__['_new_allowed'] = new_allowed
        for k, v in (init_dict or {}).items():
            if isinstance(v, dict) and not isinstance(v, CfgNode):
                v = type(self)(v)
            dict.__setitem__(self, k, v)

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.__dict__['_frozen']:
            raise AttributeError('frozen')
        self[name] = value

    def clone(self):
        return copy.deepcopy(self)

    def freeze(self):
        self.__dict__['_frozen'] = True
        for v in self.values():
            if isinstance(v, CfgNode):
                v.freeze()

    def defrost(self):
        self.__dict__['_frozen'] = False
        for v in self.values():
            if isinstance(v, CfgNode):
                v.defrost()

    def set_new_allowed(self, flag):
        self.__dict__['_new_allowed'] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v.set_new_allowed(flag)

    def _merge(self, other, path=()):
        for k, v in other.items():
            if k not in self:
                if self.__dict__['_new_allowed']:
                    dict.__setitem__(self, k, copy.deepcopy(v))
                    continue
                raise KeyError('Non-existent config key: {}'.format('.'.join(path + (k,))))
            if isinstance(self[k], CfgNode) and isinstance(v, dict):
                self[k]._merge(v, path + (k,))
            else:
                old = self[k]
                if isinstance(old, tuple) and isinstance(v, list):
                    v = tuple(v)
                elif isinstance(old, list) and isinstance(v, tuple):
                    v = list(v)
                dict.__setitem__(self, k, copy.deepcopy(v))

    def merge_from_other_cfg(self, other):
        self._merge(other)

    def merge_from_file(self, path):
        with open(path) as f:
            self._merge(type(self)(yaml.safe_load(f)))

    def merge_from_list(self, opts):
        assert len(opts) % 2 == 0
        for k, v in zip(opts[0::2], opts[1::2]):
            node = self
            parts = k.split('.')
            for p in parts[:-1]:
                node = node[p]
            try:
                v = yaml.safe_load(v)
            except Exception:
                pass
            dict.__setitem__(node, parts[-1], v)
