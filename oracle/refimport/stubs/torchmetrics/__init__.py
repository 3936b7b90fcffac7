class JaccardIndex:
    def __init__(self, *a, **k):
        pass
