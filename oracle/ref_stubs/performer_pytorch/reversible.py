class ReversibleSequence:  # only used by the whole-model ``Performer`` class
    def __init__(self, *a, **k):
        raise NotImplementedError


class SequentialSequence(ReversibleSequence):
    pass
