class _ResolutionFailures:
    pass
