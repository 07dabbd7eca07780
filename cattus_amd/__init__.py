"""cattus_amd -- MI355X-native self-play rollout path for Cattus (leaf evaluation on HIP)."""

import os as _os

# Kernel arguments in device memory: the HIP runtime reads this once, when it initialises (first HIP call of the
# process, whoever makes it), so it has to be in the environment before that; see cattus_amd/csrc/evaluator.hip.  The
# native library itself never touches the environment; a host that does not want the package to either sets
# CATTUS_NO_ENV_DEFAULTS=1 (an explicit HIP_FORCE_DEV_KERNARG always wins).
if _os.environ.get("CATTUS_NO_ENV_DEFAULTS") != "1":
    _os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

__version__ = "0.1.0"
