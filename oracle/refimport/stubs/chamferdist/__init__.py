class ChamferDistance:
    def __init__(self, *a, **k):
        pass
