class LocalAttention:  # never instantiated on the GPS path (local_heads == 0)
    def __init__(self, *a, **k):
        raise NotImplementedError
