"""cattus_amd -- MI355X-native self-play rollout path for Cattus (leaf evaluation on HIP)."""

__version__ = "0.1.0"
