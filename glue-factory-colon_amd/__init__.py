"""glue_factory_colon_amd: MI355X-native SuperPoint + LightGlue hot path.

A from-scratch gfx950 implementation of glue-factory's feature-extraction + matching
path, selected through the reference's own model registry by dotted name, e.g.

    model.extractor.name = glue_factory_colon_amd.superpoint_open
    model.matcher.name   = glue_factory_colon_amd.lightglue

(`gluefactory.models.get_model` tries the bare dotted path first,
gluefactory/models/__init__.py:7-30).  The arithmetic runs in hand-written HIP kernels
behind the C ABI declared in include/gfc_amd.h; this package is the Python host side
(tensor plumbing only) and fails loudly when the HIP library is missing.
"""
__version__ = "0.1.0"
