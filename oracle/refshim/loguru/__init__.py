class _Logger:
    def __getattr__(self, name):
        def _noop(*a, **k):
            return None
        return _noop


logger = _Logger()
