class Registry:
    """name -> builder map (same call convention as fvcore Registry: `.register()` decorator, `.get(name)`)."""

    def __init__(self, name):
        self._name = name
        self._map = {}

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self._do(o.__name__, o)
                return o
            return deco
        self._do(obj.__name__, obj)
        return obj

    def _do(self, name, obj):
        if name in self._map:
            raise KeyError("'%s' already registered in %s" % (name, self._name))
        self._map[name] = obj

    def get(self, name):
        if name not in self._map:
            raise KeyError("No object named '%s' found in '%s' registry!" % (name, self._name))
        return self._map[name]

    def __contains__(self, name):
        return name in self._map

    def keys(self):
        return self._map.keys()


META_ARCH_REGISTRY = Registry("META_ARCH")
BACKBONE_REGISTRY = Registry("BACKBONE")
PROPOSAL_GENERATOR_REGISTRY = Registry("PROPOSAL_GENERATOR")
ROI_HEADS_REGISTRY = Registry("ROI_HEADS")
