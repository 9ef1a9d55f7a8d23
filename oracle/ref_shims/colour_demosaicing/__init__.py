def demosaicing_CFA_Bayer_bilinear(*a, **k):
    raise NotImplementedError("colour_demosaicing stub")
