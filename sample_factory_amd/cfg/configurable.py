"""`Configurable` under the reference's module path (sample_factory/cfg/configurable.py:4-6): anything that keeps the
experiment configuration as `self.cfg` (model modules, learners)."""


class Configurable:
    def __init__(self, cfg):
        self.cfg = cfg
