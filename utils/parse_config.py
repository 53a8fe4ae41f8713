from vtc_amd.host.parse_config import ConfigParser, read_jsonc  # noqa: F401
