"""`from modules.utils.set_chinese_font import set_chinese_font` (reference scripts/2_feature_extraction.py:22,
scripts/3_classification.py:16): the reference's function points Matplotlib at a CJK font file for its figures.  Plotting is
outside this path and the mirrors draw nothing, so this only keeps the scripts' import line resolving."""


def set_chinese_font():
    """Nothing to configure: no figure is drawn by this implementation."""
    return None
