"""ctypes binding of libswingnn_hip.so (include/swingnn_hip.h).

The product path has no fallback: if the shared library is missing or does not load, importing
this module raises, and so does every op.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SWG_LIB: another build of the same library (A/B timing of kernel variants in scripts/); never a different backend
LIB_PATH = os.environ.get("SWG_LIB") or os.path.join(_HERE, "libswingnn_hip.so")

SWG_NOISE_EMB = 512
ERRORS = {-1: "SWG_E_SHAPE", -2: "SWG_E_UNSUPPORTED", -3: "SWG_E_WORKSPACE", -4: "SWG_E_NULL"}

_fp = C.c_void_p  # device pointers travel as void*


class SwgLin(C.Structure):
    _fields_ = [("w", _fp), ("b", _fp), ("N", C.c_int32), ("K", C.c_int32), ("Kp", C.c_int32)]


class SwgLn(C.Structure):
    _fields_ = [("g", _fp), ("b", _fp)]


class SwgBlock(C.Structure):
    _fields_ = [("C", C.c_int32), ("res", C.c_int32), ("win", C.c_int32), ("shift", C.c_int32),
                ("heads", C.c_int32), ("cond_off", C.c_int32),
                ("ln1", SwgLn), ("ln2", SwgLn),
                ("qkv", SwgLin), ("proj", SwgLin), ("fc1", SwgLin), ("fc2", SwgLin),
                ("rpb", _fp), ("mlp_img", _fp), ("attn_img", _fp), ("proj_mlp_img", _fp), ("rpb_tiles", _fp)]


class SwgMerge(C.Structure):
    _fields_ = [("C", C.c_int32), ("res", C.c_int32), ("ln", SwgLn), ("red", SwgLin), ("img", C.c_void_p)]


class SwgBreakup(C.Structure):
    _fields_ = [("Cin", C.c_int32), ("res", C.c_int32), ("pre", SwgLin), ("ln", SwgLn), ("post_ln", SwgLn),
                ("post", SwgLin), ("img_pre", C.c_void_p), ("img_post", C.c_void_p)]


class SwgLayer(C.Structure):
    _fields_ = [("n_blocks", C.c_int32), ("is_decoder", C.c_int32),
                ("blocks", C.POINTER(SwgBlock)), ("down", C.POINTER(SwgMerge)), ("up", C.POINTER(SwgBreakup))]


class SwgNet(C.Structure):
    _fields_ = [("N", C.c_int32), ("p", C.c_int32), ("in_ch", C.c_int32), ("C0", C.c_int32),
                ("n_layers", C.c_int32), ("self_cond", C.c_int32), ("cond_total", C.c_int32),
                ("pe_cond_off", C.c_int32),
                ("freqs", _fp), ("map0_w", _fp), ("map0_b", _fp), ("map1_w", _fp), ("map1_b", _fp),
                ("aff_w", _fp), ("aff_b", _fp),
                ("pe_w", _fp), ("pe_b", _fp), ("pe_ln", SwgLn),
                ("layers", C.POINTER(SwgLayer)),
                ("norm", SwgLn), ("ro_w", _fp), ("ro_b", C.c_float)]


class SwgNodeAdj(C.Structure):
    _fields_ = [("Ca", C.c_int32), ("Cn", C.c_int32), ("ro", SwgLin), ("adj_fc1", SwgLin), ("node_fc1", SwgLin),
                ("adj_fc2_w", _fp), ("adj_fc2_b", _fp), ("node_fc2_w", _fp), ("node_fc2_b", _fp)]


class SwgLinTrain(C.Structure):
    _fields_ = [("wt", _fp), ("dw", _fp), ("db", _fp)]


class SwgLnTrain(C.Structure):
    _fields_ = [("dg", _fp), ("db", _fp)]


class SwgBlockTrain(C.Structure):
    _fields_ = [("ln1", SwgLnTrain), ("ln2", SwgLnTrain), ("qkv", SwgLinTrain), ("proj", SwgLinTrain),
                ("fc1", SwgLinTrain), ("fc2", SwgLinTrain), ("drpb", _fp), ("mlp_bwd_img", _fp)]


class SwgMergeTrain(C.Structure):
    _fields_ = [("ln", SwgLnTrain), ("red", SwgLinTrain)]


class SwgBreakupTrain(C.Structure):
    _fields_ = [("pre", SwgLinTrain), ("ln", SwgLnTrain), ("post_ln", SwgLnTrain), ("post", SwgLinTrain)]


class SwgEndsTrain(C.Structure):
    _fields_ = [("pe_wb", _fp), ("dpe_w", _fp), ("dpe_b", _fp), ("pe_ln", SwgLnTrain), ("norm", SwgLnTrain),
                ("ro_wb", _fp), ("dro_w", _fp), ("dro_b", _fp)]


class SwgNodeAdjTrain(C.Structure):
    _fields_ = [("ro_wt", _fp), ("dro_w", _fp), ("dro_b", _fp), ("adj_fc1", SwgLinTrain), ("node_fc1", SwgLinTrain),
                ("dadj_fc2_w", _fp), ("dadj_fc2_b", _fp), ("dnode_fc2_w", _fp), ("dnode_fc2_b", _fp),
                ("norm", SwgLnTrain)]


class SwgRefreshItem(C.Structure):
    _fields_ = [("src0", _fp), ("src1", _fp), ("dst", _fp), ("idx", _fp), ("mode", C.c_int32), ("n0", C.c_int32),
                ("N", C.c_int32), ("K", C.c_int32), ("ld", C.c_int32), ("pad_", C.c_int32), ("count", C.c_int64)]


# every symbol include/swingnn_hip.h declares: name -> (restype, argtypes)
_i32, _sz, _f = C.c_int32, C.c_size_t, C.c_float
SYMBOLS = {
    "swg_version": (C.c_int, []),
    "swg_workspace_bytes": (_sz, [C.POINTER(SwgNet), _i32]),
    "swg_fused_op_count": (_i32, [C.POINTER(SwgNet), _i32]),
    "swg_rpb_tiles": (C.c_int, [_fp, _i32, _i32, _fp, _fp]),
    "swg_precond_forward": (C.c_int, [C.POINTER(SwgNet), _i32, _i32, _fp, _fp, _fp, _i32, _fp, _fp, _fp, _fp, _sz, _fp]),
    "swg_precond_forward_cond": (C.c_int, [C.POINTER(SwgNet), _i32, _i32, _fp, _fp, _fp, _i32, _fp, _fp, _fp, _fp, _fp, _sz, _fp]),
    "swg_cond_table": (C.c_int, [C.POINTER(SwgNet), _i32, _i32, _fp, _fp, _fp, _sz, _fp]),
    "swg_noise_embed": (C.c_int, [C.POINTER(SwgNet), _i32, _fp, _fp, _fp, _sz, _fp]),
    "swg_patch_embed_fwd": (C.c_int, [C.POINTER(SwgNet), _i32, _fp, _fp, _fp, _i32, _fp, _i32, _fp, _fp]),
    "swg_wattn_fwd": (C.c_int, [C.POINTER(SwgBlock), _i32, _fp, _fp, _i32, _fp, _fp, _sz, _fp]),
    "swg_block_fwd": (C.c_int, [C.POINTER(SwgBlock), _i32, _fp, _fp, _i32, _fp, _fp, _sz, _fp]),
    "swg_ln_mlp_fwd": (C.c_int, [C.POINTER(SwgBlock), _i32, _fp, _fp, _fp, _sz, _fp]),
    "swg_patch_merge_fwd": (C.c_int, [C.POINTER(SwgMerge), _i32, _fp, _fp, _fp, _sz, _fp]),
    "swg_patch_expand_fwd": (C.c_int, [C.POINTER(SwgBreakup), _i32, _fp, _fp, _fp, _fp, _sz, _fp]),
    "swg_readout_sym_precond_fwd": (C.c_int, [C.POINTER(SwgNet), _i32, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _sz, _fp]),
    "swg_linear_bf16": (C.c_int, [_i32, _i32, _i32, _fp, _i32, _fp, _fp, _i32, _fp, _fp, _fp, _i32, _fp]),
    "swg_linear_wgrad_workspace_bytes": (_sz, [_i32, _i32]),
    "swg_linear_wgrad_bf16": (C.c_int, [_i32, _i32, _i32, _fp, _i32, _fp, _i32, _fp, _fp, _sz, _fp]),
    "swg_heun_churn": (C.c_int, [_i32, _i32, _fp, _fp, _fp, _f, _fp, _fp]),
    "swg_heun_euler": (C.c_int, [_i32, _i32, _fp, _fp, _fp, _f, _f, _i32, _fp, _fp, _fp]),
    "swg_heun_correct": (C.c_int, [_i32, _i32, _fp, _fp, _fp, _fp, _fp, _f, _f, _fp, _fp]),
    "swg_mask_adjs": (C.c_int, [_i32, _i32, _fp, _fp, _fp]),
    "swg_sym_check": (C.c_int, [_i32, _i32, _fp, _fp, _fp]),
    "swg_nodeadj_workspace_bytes": (_sz, [C.POINTER(SwgNet), C.POINTER(SwgNodeAdj), _i32]),
    "swg_nodeadj_forward": (C.c_int, [C.POINTER(SwgNet), C.POINTER(SwgNodeAdj), _i32, _i32, _fp, _fp, _fp, _fp, _i32, _fp,
                                      _fp, _fp, _fp, _fp, _fp, _sz, _fp]),
    "swg_heun_nodes_churn": (C.c_int, [_i32, _i32, _i32, _fp, _fp, _fp, _f, _fp, _fp]),
    "swg_heun_nodes_euler": (C.c_int, [_i32, _i32, _i32, _fp, _fp, _fp, _f, _f, _i32, _fp, _fp, _fp]),
    "swg_heun_nodes_correct": (C.c_int, [_i32, _i32, _i32, _fp, _fp, _fp, _fp, _fp, _f, _f, _fp, _fp]),
    "swg_edm_objective": (C.c_int, [_i32, _i32, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
    "swg_edm_loss": (C.c_int, [_i32, _i32, _fp, _fp, _fp, _fp, _f, _i32, _fp, _fp, _fp]),
    "swg_edm_loss_noalloc": (C.c_int, [_i32, _i32, _fp, _fp, _fp, _fp, _f, _i32, _fp, _fp]),
    "swg_edm_loss_noalloc_warmup": (C.c_int, []),
    "swg_train_workspace_bytes": (_sz, [C.POINTER(SwgNet), _i32]),
    "swg_edm_coef": (C.c_int, [_i32, _fp, _i32, _i32, _fp, _fp, _fp]),
    "swg_wattn_bwd": (C.c_int, [C.POINTER(SwgBlock), C.POINTER(SwgBlockTrain), _i32, _fp, _fp, _i32, _fp, _fp, _fp, _i32,
                                _fp, _sz, _fp]),
    "swg_ln_mlp_bwd": (C.c_int, [C.POINTER(SwgBlock), C.POINTER(SwgBlockTrain), _i32, _fp, _fp, _fp, _fp, _sz, _fp]),
    "swg_patch_merge_bwd": (C.c_int, [C.POINTER(SwgMerge), C.POINTER(SwgMergeTrain), _i32, _fp, _fp, _fp, _fp, _sz, _fp]),
    "swg_patch_expand_bwd": (C.c_int, [C.POINTER(SwgBreakup), C.POINTER(SwgBreakupTrain), _i32, _fp, _fp, _fp, _fp, _fp,
                                       _fp, _sz, _fp]),
    "swg_patch_embed_bwd": (C.c_int, [C.POINTER(SwgNet), C.POINTER(SwgEndsTrain), _i32, _fp, _fp, _fp, _i32, _fp, _i32,
                                      _fp, _fp, _i32, _fp, _sz, _fp]),
    "swg_readout_sym_precond_bwd": (C.c_int, [C.POINTER(SwgNet), C.POINTER(SwgEndsTrain), _i32, _fp, _fp, _fp, _fp, _fp,
                                              _fp, _sz, _fp]),
    "swg_edm_loss_bwd": (C.c_int, [_i32, _i32, _fp, _fp, _fp, _fp, _fp, _f, _fp, _fp]),
    "swg_nodeadj_train_workspace_bytes": (_sz, [C.POINTER(SwgNet), C.POINTER(SwgNodeAdj), _i32]),
    "swg_nodeadj_embed_fwd": (C.c_int, [C.POINTER(SwgNet), C.POINTER(SwgNodeAdj), _i32, _fp, _fp, _fp, _fp, _fp, _fp, _i32,
                                        _fp, _i32, _fp, _fp]),
    "swg_nodeadj_readout_fwd": (C.c_int, [C.POINTER(SwgNet), C.POINTER(SwgNodeAdj), _i32, _fp, _fp, _fp, _fp, _fp, _fp,
                                          _fp, _fp, _fp, _sz, _fp]),
    "swg_nodeadj_embed_bwd": (C.c_int, [C.POINTER(SwgNet), C.POINTER(SwgNodeAdj), C.POINTER(SwgEndsTrain), _i32, _fp, _fp,
                                        _fp, _fp, _fp, _fp, _i32, _fp, _i32, _fp, _fp, _i32, _fp, _sz, _fp]),
    "swg_nodeadj_readout_bwd": (C.c_int, [C.POINTER(SwgNet), C.POINTER(SwgNodeAdj), C.POINTER(SwgNodeAdjTrain), _i32, _fp,
                                          _fp, _fp, _fp, _fp, _fp, _fp, _sz, _fp]),
    "swg_nodes_objective": (C.c_int, [_i32, _i32, _i32, _fp, _fp, _fp, _fp, _fp, _fp]),
    "swg_nodes_loss": (C.c_int, [_i32, _i32, _i32, _fp, _fp, _fp, _fp, _f, _fp, _fp]),
    "swg_nodes_loss_bwd": (C.c_int, [_i32, _i32, _i32, _fp, _fp, _fp, _fp, _fp, _f, _fp, _fp]),
    "swg_optim_chunk": (C.c_int, []),
    "swg_grad_sqnorm": (C.c_int, [_fp, _fp, _fp, _fp, _i32, _fp, _fp]),
    "swg_adam_ema_step": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _fp, _i32, _fp, _fp, _fp, C.c_int64, _i32,
                                    C.POINTER(C.c_float), _fp, _f, _f, _f, _f, _f, _f, _i32, _fp]),
    "swg_attr_decode": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _fp, _fp, _fp, _fp, _fp, _fp]),
    "swg_attr_encode": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _i32, _i32, _fp, _fp, _fp, _fp]),
    "swg_graph_stats_workspace_bytes": (_sz, [_i32, _i32]),
    "swg_graph_hists": (C.c_int, [_i32, _i32, _i32, _fp, _fp, _fp, _fp, C.c_double, _fp, _sz, _fp]),
    "swg_hist_pmf": (C.c_int, [_i32, _i32, _i32, _fp, C.c_double, C.c_double, C.c_double, _fp, _fp]),
    "swg_mmd_gaussian_tv": (C.c_int, [_i32, _i32, _i32, _fp, _fp, C.c_double, _fp, _fp, _fp]),
    "swg_refresh_derived": (C.c_int, [_fp, _fp, _fp, _fp, _i32, _fp]),
    "swg_prof_enable": (C.c_int, [_i32]),
    "swg_prof_reset": (C.c_int, []),
    "swg_prof_kernels": (C.c_int, []),
    "swg_prof_name": (C.c_char_p, [_i32]),
    "swg_prof_read": (C.c_int, [_i32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                C.POINTER(C.c_int64)]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C swingnn_amd/csrc` (or __graft_entry__.build()). "
            "swingnn_amd has no CPU or PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


class SwgError(RuntimeError):
    pass


def check(code: int, what: str):
    if code == 0:
        return
    if code < 0:
        raise SwgError(f"{what}: {ERRORS.get(code, code)}")
    raise SwgError(f"{what}: hipError_t {code}")
