"""Stand-in for the handful of MONAI 1.1.0 helper symbols the reference imports.

TEST/FIXTURE TOOLING ONLY.  MONAI is not installed in the build image, and every
file under /root/reference/networks except norms/ imports it at module load
(e.g. reference networks/nets/swin_unetr.py:20).  This package is put on
sys.path *only* by oracle/tools/make_golden.py so that the reference's own
modules can be imported and run on CPU to produce golden vectors.  The product
never imports it.  Anything whose arithmetic lives here rather than in the
reference (MLPBlock, SABlock, DropPath) is restated from MONAI 1.1.0's public
API and is flagged "unpinned by the reference" in the fixture metadata.
Four symbols delegate to the reference's own vendored copies (factories,
get_act_layer, Convolution, PatchEmbeddingBlock) -- see SURVEY.md section 8(c).
"""
