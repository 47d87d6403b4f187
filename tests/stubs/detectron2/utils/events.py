"""[D2-upstream] detectron2.utils.events.get_event_storage / EventStorage, the part the ROI heads use (test stub)."""
_CURRENT = []


class EventStorage:
    def __init__(self):
        self.scalars = {}

    def put_scalar(self, name, value, smoothing_hint=True):
        self.scalars[name] = float(value)

    def __enter__(self):
        _CURRENT.append(self)
        return self

    def __exit__(self, *exc):
        assert _CURRENT[-1] is self
        _CURRENT.pop()


def get_event_storage():
    assert len(_CURRENT), "get_event_storage() has to be called inside a 'with EventStorage(...)' context!"
    return _CURRENT[-1]
