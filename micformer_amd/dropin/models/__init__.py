"""Drop-in `models` package: the reference's module names (`MicFormer/models/MICFormer_self.py`, `STN.py`) on the HIP modules.

    sys.path.insert(0, "<repo>/micformer_amd/dropin")
    from models.MICFormer_self import Head          # MicFormer/test.ipynb:11, the intent of train_mmwhs_noPad.py:26
"""
