"""MI355X-native CLIP-conditioned diffusion-LM captioning hot path.

The directory name carries a hyphen (it mirrors the reference repo's name), so import it with
`importlib.import_module("diffusion-image-captioning_amd")`; the first import registers the
alias `dic_amd` so `import dic_amd` works afterwards.
"""
import sys as _sys

_sys.modules.setdefault("dic_amd", _sys.modules[__name__])
