def colored(s, *a, **k):
    return s
