"""microclimf_amd — MI355X-native grid microclimate solver (runmicro1Cpp /
runmicro2Cpp hot path of ilyamaclean/microclimf) behind a C ABI."""
from ._abi import McfError, OUT_NAMES  # noqa: F401
