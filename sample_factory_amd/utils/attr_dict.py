"""Dict with attribute access (mirrors sample_factory/utils/attr_dict.py semantics: cfg objects may be AttrDict or Namespace)."""


class AttrDict(dict):
    __setattr__ = dict.__setitem__

    def __getattr__(self, attr):
        try:
            return self[attr]
        except KeyError as e:
            raise AttributeError(attr) from e

    def __delattr__(self, attr):
        del self[attr]
