"""ctypes binding of libvnr_amd.so (include/vnr_amd.h).  No CPU fallback: if the library is missing
or no gfx950 device is present, calls fail loudly."""
import ctypes as C
import os
import re
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
SO_PATH = os.environ.get("VNR_AMD_LIB_PATH") or os.path.join(_HERE, "libvnr_amd.so")   # the override is for A/B builds of kernel variants (tools/ab_build.sh)
HEADER = os.path.join(ROOT, "include", "vnr_amd.h")

_lib = None
_torch_imported_before_load = None  # set when the library is loaded


class VnrAmdError(RuntimeError):
    pass


def require_torch_loaded_first():
    """PyTorch-ROCm wheels bundle their own ROCm runtime (torch/lib/libamdhip64.so, libhsa-runtime64.so) while
    libvnr_amd.so links against the system ROCm.  If torch is imported first, the library binds to torch's copy and both
    share one runtime.  If the library is loaded first, `import torch` maps a SECOND HSA runtime into the process, which
    finds no GPU ("RuntimeError: No HIP GPUs are available"; measured on MI355X, ROCm 7.2 + torch 2.10+rocm7.0).  Anything
    here that uses torch on the GPU calls this, so that the wrong order fails with an explanation."""
    if _lib is not None and not _torch_imported_before_load:
        raise VnrAmdError("libvnr_amd.so was loaded before `import torch`: PyTorch's bundled ROCm runtime cannot share the "
                          "GPU with the system runtime the library already initialised.  Import torch first (instantvnr_amd."
                          "dist.init_from_env does when WORLD_SIZE > 1; tests/conftest.py does for the test session).")


def build(force=False):
    """compile libvnr_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)"""
    args = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8", "-s"]
    if force:
        subprocess.check_call(args + ["clean"])
    subprocess.check_call(args)
    return SO_PATH


class FrameStats(C.Structure):
    _fields_ = [("n_samples", C.c_uint64), ("n_reference_slots", C.c_uint64), ("n_iterations", C.c_uint32),
                ("n_rays_hit", C.c_uint32), ("infer_kernel_ms", C.c_double), ("infer_kernel_launches", C.c_uint64), ("infer_union_ms", C.c_double)]


class OutOfCoreInfo(C.Structure):
    _fields_ = [("file_dims", C.c_int * 3), ("block_dims", C.c_int * 3), ("block_index_space", C.c_int * 3),
                ("n_blocks", C.c_uint64), ("n_concurrent_blocks", C.c_uint64), ("block_size_aligned", C.c_uint64),
                ("bytes_read", C.c_uint64)]


def declared_symbols():
    """every function name declared in include/vnr_amd.h"""
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vnrAmd[A-Za-z0-9_]+)\s*\(", text)))


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise VnrAmdError(f"{SO_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)")
    global _torch_imported_before_load
    _torch_imported_before_load = "torch" in sys.modules
    L = C.CDLL(SO_PATH)
    P, I, U32, U64, F, D, SZ = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_float, C.c_double, C.c_size_t
    FP = C.POINTER(C.c_float)
    IP = C.POINTER(C.c_int)

    def sig(name, res, *args):
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = list(args)

    sig("vnrAmdGetLastError", C.c_char_p)
    sig("vnrAmdBuildId", C.c_char_p)
    sig("vnrAmdVersion", C.c_char_p)
    sig("vnrAmdInit", I, I)
    sig("vnrAmdDeviceCount", I)
    sig("vnrAmdHasDevice", I)
    sig("vnrAmdMalloc", P, SZ)
    sig("vnrAmdFree", I, P)
    sig("vnrAmdMemcpyH2D", I, P, P, SZ)
    sig("vnrAmdMemcpyD2H", I, P, P, SZ)
    sig("vnrAmdMemset", I, P, I, SZ)
    sig("vnrAmdSynchronize", I)
    sig("vnrAmdDefaultStream", P)
    sig("vnrAmdJsonConvert", I, P, SZ, I, I, C.POINTER(P), C.POINTER(SZ))
    sig("vnrAmdJsonSave", I, P, SZ, I, C.c_char_p, I)
    sig("vnrAmdFreeHost", None, P)
    sig("vnrAmdCreateCamera", P)
    sig("vnrAmdCameraSet", I, P, FP, FP, FP)
    sig("vnrAmdCameraSetFovy", I, P, F)
    sig("vnrAmdCameraGet", I, P, FP, FP, FP, FP)
    sig("vnrAmdReleaseCamera", None, P)
    sig("vnrAmdCreateTransferFunction", P)
    sig("vnrAmdTransferFunctionSetColor", I, P, FP, I)
    sig("vnrAmdTransferFunctionSetAlpha", I, P, FP, I)
    sig("vnrAmdTransferFunctionSetValueRange", I, P, F, F)
    sig("vnrAmdTransferFunctionGetSizes", I, P, IP, IP)
    sig("vnrAmdTransferFunctionGet", I, P, FP, FP, FP)
    sig("vnrAmdReleaseTransferFunction", None, P)
    sig("vnrAmdCreateSimpleVolumeFromMemory", P, P, IP, I, F, F)
    sig("vnrAmdCreateSimpleVolumeFromRawFile", P, C.c_char_p, IP, I, SZ, I, F, F)
    sig("vnrAmdCreateSimpleVolumePerlin", P, IP, U32, I, F)
    sig("vnrAmdSimpleVolumeDeviceData", P, P)
    sig("vnrAmdCreateSimpleVolumeOutOfCore", P, C.c_char_p, IP, I, SZ, F, F, U64, U64)
    sig("vnrAmdCreateSimpleVolumeFromScene", P, P, SZ, I, C.c_char_p, I)
    sig("vnrAmdSimpleVolumeGetNumberOfTimeSteps", I, P)
    sig("vnrAmdSimpleVolumeSetCurrentTimeStep", I, P, I)
    sig("vnrAmdSceneGetValueRange", I, P, SZ, I, FP)
    sig("vnrAmdCameraSetFromScene", I, P, P, SZ, I)
    sig("vnrAmdSimpleVolumeOutOfCoreInfo", I, P, C.POINTER(OutOfCoreInfo))
    sig("vnrAmdSimpleVolumeOutOfCoreBlocks", I, P, IP, SZ)
    sig("vnrAmdSimpleVolumeOutOfCoreSetAsyncRefresh", I, P, I)
    sig("vnrAmdSimpleVolumeOutOfCoreRefreshStats", I, P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))
    sig("vnrAmdCreateNeuralVolume", P, P, SZ, I, P, I)
    sig("vnrAmdCreateNeuralVolumeFromDims", P, P, SZ, I, IP)
    sig("vnrAmdCreateNeuralVolumeFromParams", P, P, SZ, I)
    sig("vnrAmdNeuralVolumeSetModel", I, P, P, SZ, I)
    sig("vnrAmdNeuralVolumeSetParams", I, P, P, SZ, I)
    sig("vnrAmdNeuralVolumeGetPSNR", D, P, I)
    sig("vnrAmdNeuralVolumeGetSSIM", D, P, I)
    sig("vnrAmdNeuralVolumeGetTestingLoss", D, P)
    sig("vnrAmdNeuralVolumeGetTrainingLoss", D, P)
    sig("vnrAmdNeuralVolumeGetTrainingStep", I, P)
    sig("vnrAmdNeuralVolumeGetNumberOfBlobs", I, P)
    sig("vnrAmdNeuralVolumeTrain", I, P, I, I)
    sig("vnrAmdNeuralVolumeSerializeParamsToFile", I, P, C.c_char_p)
    sig("vnrAmdNeuralVolumeDecodeProgressive", I, P)
    sig("vnrAmdNeuralVolumeDecodeInference", I, P, C.c_char_p)
    sig("vnrAmdNeuralVolumeDecodeReference", I, P, C.c_char_p)
    sig("vnrAmdNeuralVolumeDecodedDeviceData", P, P)
    sig("vnrAmdNeuralVolumeSerializeParams", I, P, C.POINTER(P), C.POINTER(SZ))
    sig("vnrAmdNeuralVolumeInference", I, P, SZ, P, P, P)
    sig("vnrAmdNeuralVolumeEncode", I, P, SZ, P, P, P)
    sig("vnrAmdNeuralVolumeBrickImageInfo", I, P, IP, C.POINTER(SZ), FP)
    sig("vnrAmdNeuralVolumeGetInfo", I, P, IP, IP, IP, IP, IP, C.POINTER(U64))
    sig("vnrAmdNeuralVolumeGetModelKind", I, P, IP, IP, IP, IP, IP, IP)
    sig("vnrAmdNeuralVolumeLevelTable", I, P, I, P, P, P, P)
    sig("vnrAmdNeuralVolumeGetParamsFP16", I, P, P, SZ)
    sig("vnrAmdNeuralVolumeSetParamsFP16", I, P, P, SZ)
    sig("vnrAmdNeuralVolumeTrainBegin", I, P)
    sig("vnrAmdNeuralVolumeGradients", P, P, C.POINTER(SZ))
    sig("vnrAmdNeuralVolumeTrainEnd", I, P, F, I)
    sig("vnrAmdNeuralVolumeForwardBackward", I, P, SZ, P, P)
    sig("vnrAmdNeuralVolumeTrainingBuffer", I, P, I, C.POINTER(P), C.POINTER(SZ))
    sig("vnrAmdNeuralVolumeRescatterGridGradients", I, P, SZ, P)
    sig("vnrAmdNeuralVolumeGradientDistance", I, P, P, C.POINTER(C.c_double))
    sig("vnrAmdNeuralVolumeSetSamplerSeed", I, P, U64, U64)
    sig("vnrAmdNeuralVolumeSetInitSeed", I, P, U64)
    sig("vnrAmdVolumeSetClippingBox", I, P, FP, FP)
    sig("vnrAmdVolumeSetScaling", I, P, FP)
    sig("vnrAmdVolumeSetTransform", I, P, FP)
    sig("vnrAmdSimpleVolumeGetDataRange", I, P, FP)
    sig("vnrAmdMarchingCube", I, P, F, C.POINTER(C.POINTER(C.c_float)), C.POINTER(SZ), I)
    sig("vnrAmdSaveTriangles", I, C.c_char_p, C.POINTER(C.c_float), SZ)
    sig("vnrAmdVolumeGetValueRange", I, P, FP)
    sig("vnrAmdVolumeGetDims", I, P, IP)
    sig("vnrAmdVolumeIsNetwork", I, P)
    sig("vnrAmdVolumeGetMacrocell", I, P, IP, FP, C.POINTER(P), C.POINTER(P))
    sig("vnrAmdReleaseVolume", None, P)
    sig("vnrAmdCreateRenderer", P, P)
    sig("vnrAmdRendererSetFramebufferSize", I, P, I, I)
    sig("vnrAmdRendererSetTransferFunction", I, P, P)
    sig("vnrAmdRendererSetCamera", I, P, P)
    sig("vnrAmdRendererSetMode", I, P, I)
    sig("vnrAmdRendererSetDenoiser", I, P, I)
    sig("vnrAmdRendererSetVolumeSamplingRate", I, P, F)
    sig("vnrAmdRendererSetVolumeDensityScale", I, P, F)
    sig("vnrAmdRendererResetAccumulation", I, P)
    sig("vnrAmdRender", I, P)
    sig("vnrAmdRendererMapFrame", P, P)
    sig("vnrAmdRendererSetOutputAsDeviceFramebuffer", I, P, I)
    sig("vnrAmdRendererSetPixelRange", I, P, U32, U32)
    sig("vnrAmdRendererSetPixelInterleave", I, P, U32, U32, U32)
    sig("vnrAmdRendererGetFrameStats", I, P, C.POINTER(FrameStats))
    sig("vnrAmdRendererSetProfiling", I, P, I)
    sig("vnrAmdRendererSetAsync", I, P, I)
    sig("vnrAmdRendererSetInShaderKernel", I, P, I)
    sig("vnrAmdRendererDebugQueues", I, P, C.POINTER(P), C.POINTER(P), FP, I)
    sig("vnrAmdRendererDebugSchedule", I, P, IP)
    sig("vnrAmdReleaseRenderer", None, P)
    sig("vnrAmdDistGetUniqueId", I, P)
    sig("vnrAmdDistInit", I, I, I, I, P, C.c_char_p, C.c_char_p)
    sig("vnrAmdDistInitFromEnv", I)
    sig("vnrAmdDistFinalize", I)
    sig("vnrAmdDistRank", I)
    sig("vnrAmdDistWorldSize", I)
    sig("vnrAmdDistTransport", C.c_char_p)
    sig("vnrAmdDistRcclRanksSeen", I)
    sig("vnrAmdDistBarrier", I)
    sig("vnrAmdDistAllReduceHost", I, C.POINTER(D), I, I)
    sig("vnrAmdDistAllReduce", I, P, SZ, I, I)
    sig("vnrAmdDistAllGather", I, P, P, SZ)
    sig("vnrAmdDistReduceScatter", I, P, SZ, I)
    sig("vnrAmdDistBroadcast", I, P, SZ, I)
    sig("vnrAmdDistSelfTest", I, C.c_double, C.c_char_p, SZ)
    sig("vnrAmdRendererSetDistributed", I, P, I)
    sig("vnrAmdRendererGatherFrame", P, P)
    sig("vnrAmdRendererRenderPipelined", I, P, C.POINTER(P))
    sig("vnrAmdRendererGetCompletedFrameStats", I, P, C.POINTER(FrameStats))
    sig("vnrAmdRendererFlushPipeline", I, P, C.POINTER(P))
    sig("vnrAmdNeuralVolumeTrainDataParallel", I, P, I, I)
    sig("vnrAmdNeuralVolumeSyncReplicas", I, P)
    sig("vnrAmdNeuralVolumeSetBrickImageMode", I, P, I)
    sig("vnrAmdNeuralVolumeSetBrickImageBudget", I, P, SZ)
    sig("vnrAmdNeuralVolumeBrickImageLevels", C.c_uint, P)
    sig("vnrAmdNeuralVolumeGridBackwardPlan", I, P, U64, C.POINTER(U32), C.POINTER(U64))
    sig("vnrAmdNeuralVolumeBrickImagePolicy", I, P, C.POINTER(U64), C.POINTER(C.c_uint), IP, C.POINTER(U64))
    sig("vnrAmdNeuralVolumeSetTrainProfiling", I, P, I)
    sig("vnrAmdNeuralVolumeGetTrainProfile", I, P, C.POINTER(D), IP)
    sig("vnrAmdNeuralVolumeAllReduceGradients", I, P)
    sig("vnrAmdNeuralVolumeTrainEndDataParallel", I, P, I, I)
    sig("vnrAmdNeuralVolumeSetGradients", I, P, C.POINTER(C.c_float), SZ)
    sig("vnrAmdMemoryQuery", None, C.POINTER(SZ), C.POINTER(SZ))
    sig("vnrAmdFreeTemporaryGPUMemory", None)
    sig("vnrAmdSimpleVolumeTakeSamples", I, P, SZ, FP, FP, P, P, P)
    sig("vnrAmdSimpleVolumeTakeSamplesGrid", I, P, IP, IP, P, P, P)
    sig("vnrAmdSimpleVolumeSample", I, P, SZ, P, P, I, P)
    sig("vnrAmdNeuralVolumeUpdateMacrocell", I, P, SZ, P, P, P)
    sig("vnrAmdVolumeUpdateMaxOpacity", I, P, P)
    _lib = L
    return L


def last_error():
    return lib().vnrAmdGetLastError().decode("utf-8", "replace")


def check(status):
    if status != 0:
        raise VnrAmdError(last_error())


def check_ptr(p):
    if not p:
        raise VnrAmdError(last_error())
    return p


def require_device():
    if not lib().vnrAmdHasDevice():
        raise VnrAmdError("no HIP device: the MI355X path has no CPU fallback")
