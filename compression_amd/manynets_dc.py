"""Joint denoise + compress model (mirror of src/dc/models/manynets_dc.py:36-38): the Manypriors codec fed
with noisy images and trained against clean ones, hence no visual loss of its own (lossf=None)."""
from . import manynets_compressor


class ManyPriors_DC(manynets_compressor.Balle2017ManyPriors_ImageCompressor):
    def __init__(self, **kwargs):
        kwargs.pop("lossf", None)
        super().__init__(lossf=None, **kwargs)
