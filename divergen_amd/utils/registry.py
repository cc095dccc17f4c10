"""Name -> object registry (fvcore.common.registry.Registry surface used by D2's *_REGISTRY)."""


class Registry:
    def __init__(self, name):
        self._name = name
        self._map = {}

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self._do_register(o.__name__, o)
                return o
            return deco
        self._do_register(obj.__name__, obj)
        return obj

    def _do_register(self, name, obj):
        assert name not in self._map, "'%s' already registered in %s" % (name, self._name)
        self._map[name] = obj

    def get(self, name):
        if name not in self._map:
            raise KeyError("No object named '%s' found in '%s' registry!" % (name, self._name))
        return self._map[name]

    def __contains__(self, name):
        return name in self._map

    def __iter__(self):
        return iter(self._map.items())
