"""Name -> class registries with the reference's semantics.

Mirrors the behaviour of /root/reference/lbasicsr/utils/registry.py:1-62: `register()` works as a
decorator or a call, duplicate names assert, `get()` retries with the `_lbasicsr` suffix and
raises KeyError when the name is unknown.
"""


class Registry:
    def __init__(self, name: str):
        self._name = name
        self._table = {}

    def _add(self, key, obj, suffix=None):
        if isinstance(suffix, str):
            key = f"{key}_{suffix}"
        assert key not in self._table, f"An object named '{key}' was already registered in '{self._name}' registry!"
        self._table[key] = obj

    def register(self, obj=None, suffix=None):
        if obj is not None:
            self._add(obj.__name__, obj, suffix)
            return obj

        def decorator(target):
            self._add(target.__name__, target, suffix)
            return target

        return decorator

    def get(self, name, suffix="lbasicsr"):
        found = self._table.get(name)
        if found is None:
            found = self._table.get(f"{name}_{suffix}")
        if found is None:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return found

    def __contains__(self, name):
        return name in self._table

    def __iter__(self):
        return iter(self._table.items())

    def keys(self):
        return self._table.keys()


DATASET_REGISTRY = Registry("dataset")
ARCH_REGISTRY = Registry("arch")
MODEL_REGISTRY = Registry("model")
LOSS_REGISTRY = Registry("loss")
METRIC_REGISTRY = Registry("metric")
