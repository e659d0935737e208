registry = {}


def register(id, entry_point, **kwargs):
    registry[id] = entry_point
