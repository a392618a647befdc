"""Stand-in for pyhocon on top of neuraludf_amd.conf (the HOCON subset the reference's conf files use):
ConfigFactory.parse_string / parse_file and HOCONConverter.to_hocon, as exp_runner_blending.py:38-44, 460-462 call them."""
from neuraludf_amd import conf as _conf


class ConfigFactory:
    @staticmethod
    def parse_string(text):
        return _conf.parse_string(text)

    @staticmethod
    def parse_file(path):
        return _conf.parse_string(open(path).read())


def _dump(v, ind):
    pad = "  " * ind
    if isinstance(v, dict):
        return "{\n" + "".join(f"{pad}  {k} = {_dump(x, ind + 1)}\n" for k, x in v.items()) + pad + "}"
    if isinstance(v, (list, tuple)):
        return "[" + ", ".join(_dump(x, ind) for x in v) + "]"
    if isinstance(v, bool):
        return "true" if v else "false"
    return str(v)


class HOCONConverter:
    @staticmethod
    def to_hocon(config):
        return "".join(f"{k} = {_dump(v, 0)}\n" for k, v in config.items())
