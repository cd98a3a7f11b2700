"""MI355X-native raymarch backend: HIP kernels behind a C ABI (see include/)."""
from . import api  # noqa: F401
