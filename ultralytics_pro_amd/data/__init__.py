"""Image-space steps either side of the model: `LetterBox` on the HIP path."""
from .augment import LetterBox  # noqa: F401
