"""specutils stand-in: the reference imports these names at module top but no
hot-path or setup code ever calls them."""


def vac_to_air(*a, **k):
    raise NotImplementedError('stub')


def air_to_vac(*a, **k):
    raise NotImplementedError('stub')
