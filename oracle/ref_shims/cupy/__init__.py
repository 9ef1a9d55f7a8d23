"""stub of cupy for importing the reference on CPU (see ../README.md)"""


class _Util:
    @staticmethod
    def memoize(for_each_device=False):
        def deco(fn):
            return fn
        return deco


class _Cuda:
    @staticmethod
    def compile_with_cache(src):
        raise NotImplementedError("cupy stub: CUDA kernels cannot run here")


util = _Util()
cuda = _Cuda()
