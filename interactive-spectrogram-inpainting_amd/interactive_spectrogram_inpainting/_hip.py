"""ctypes binding of libisi_hip.so (C-ABI declared in include/isi_hip.h).

PyTorch is used only as the owner of device memory and of the HIP stream;
every tensor is handed to the library as a raw device pointer.
"""
from __future__ import annotations

import ctypes as C
import os
import pathlib
import threading
from typing import Optional

import torch

_PKG_DIR = pathlib.Path(__file__).resolve().parent
_LIB_PATH = _PKG_DIR.parent / "lib" / "libisi_hip.so"

ISI_MAX_STAGES = 4
ISI_MAX_RES = 8
MODE_ENCODE, MODE_DECODE, MODE_FORWARD = 1, 2, 3


class HipLibraryError(RuntimeError):
    pass


class isi_src(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("C", C.c_int),
                ("sn", C.c_int64), ("sc", C.c_int64), ("sh", C.c_int64), ("sw", C.c_int64)]


class isi_dst(C.Structure):
    _fields_ = [("ptr", C.c_void_p),
                ("sn", C.c_int64), ("sc", C.c_int64), ("sh", C.c_int64), ("sw", C.c_int64)]


class isi_conv_w(C.Structure):
    _fields_ = [("w", C.c_void_p), ("bias", C.c_void_p), ("Cin", C.c_int), ("Cout", C.c_int)]


class isi_encoder_w(C.Structure):
    _fields_ = [("n_down", C.c_int), ("down", isi_conv_w * ISI_MAX_STAGES), ("conv3", isi_conv_w),
                ("n_res", C.c_int), ("res3", isi_conv_w * ISI_MAX_RES), ("res1", isi_conv_w * ISI_MAX_RES)]


class isi_decoder_w(C.Structure):
    _fields_ = [("conv3", isi_conv_w), ("n_res", C.c_int),
                ("res3", isi_conv_w * ISI_MAX_RES), ("res1", isi_conv_w * ISI_MAX_RES),
                ("n_up", C.c_int), ("up", isi_conv_w * ISI_MAX_STAGES)]


class isi_codebook_w(C.Structure):
    _fields_ = [("codes_kd", C.c_void_p), ("e2", C.c_void_p), ("D", C.c_int), ("K", C.c_int)]


class isi_vqvae_w(C.Structure):
    _fields_ = [("in_channel", C.c_int),
                ("enc_b", isi_encoder_w), ("enc_t", isi_encoder_w),
                ("quantize_conv_t", isi_conv_w), ("quantize_conv_b", isi_conv_w),
                ("quantize_t", isi_codebook_w), ("quantize_b", isi_codebook_w),
                ("dec_t", isi_decoder_w), ("dec", isi_decoder_w),
                ("n_upsample", C.c_int), ("upsample", isi_conv_w * ISI_MAX_STAGES), ("w16", C.c_int),
                ("precision", C.c_int), ("no_quantize", C.c_int)]


class isi_vqvae_out(C.Structure):
    _fields_ = [("dec", C.c_void_p), ("quant_t", C.c_void_p), ("quant_b", C.c_void_p),
                ("id_t", C.c_void_p), ("id_b", C.c_void_p), ("scalars", C.c_void_p)]


class isi_attn_args(C.Structure):
    _fields_ = [("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("rel_embeddings", C.c_void_p),
                ("dense_mask", C.c_void_p), ("out", C.c_void_p),
                ("Sq", C.c_int), ("Sk", C.c_int), ("B", C.c_int), ("H", C.c_int), ("head_dim", C.c_int),
                ("q_ss", C.c_int64), ("q_sb", C.c_int64), ("q_sh", C.c_int64),
                ("k_ss", C.c_int64), ("k_sb", C.c_int64), ("k_sh", C.c_int64),
                ("v_ss", C.c_int64), ("v_sb", C.c_int64), ("v_sh", C.c_int64),
                ("o_ss", C.c_int64), ("o_sb", C.c_int64), ("o_sh", C.c_int64),
                ("Cq", C.c_int), ("Ck", C.c_int), ("Ek", C.c_int), ("rel_rows", C.c_int),
                ("mask_mode", C.c_int), ("scale", C.c_float), ("lse", C.c_void_p), ("precision", C.c_int),
                ("logits", C.c_void_p), ("logits_ld", C.c_int64), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]


class isi_linear_args(C.Structure):
    _fields_ = [("x", C.c_void_p), ("ldx", C.c_int64), ("packed_w", C.c_void_p), ("bias", C.c_void_p),
                ("residual", C.c_void_p), ("ldr", C.c_int64), ("gate", C.c_void_p), ("ldg", C.c_int64),
                ("gate_scale", C.c_float), ("out", C.c_void_p), ("ldo", C.c_int64),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("flags", C.c_int),
                ("drop_p", C.c_float), ("drop_seed", C.c_uint64)]


class isi_attn_bwd_args(C.Structure):
    _fields_ = [("fwd", isi_attn_args), ("d_out", C.c_void_p), ("dq", C.c_void_p), ("dk", C.c_void_p),
                ("dv", C.c_void_p), ("d_rel", C.c_void_p), ("workspace", C.c_void_p),
                ("workspace_floats", C.c_size_t)]


ISI_MAX_LAYERS = 16


class isi_attn_w(C.Structure):
    _fields_ = [("in_proj_weight", C.c_void_p), ("in_proj_bias", C.c_void_p), ("out_proj_weight", C.c_void_p),
                ("out_proj_bias", C.c_void_p), ("rel_embeddings", C.c_void_p), ("rel_rows", C.c_int)]


class isi_decoder_layer_w(C.Structure):
    _fields_ = [("self_attn", isi_attn_w), ("cross_attn", isi_attn_w),
                ("linear1_w", C.c_void_p), ("linear1_b", C.c_void_p), ("linear2_w", C.c_void_p),
                ("linear2_b", C.c_void_p), ("norm1_w", C.c_void_p), ("norm1_b", C.c_void_p),
                ("norm2_w", C.c_void_p), ("norm2_b", C.c_void_p), ("norm3_w", C.c_void_p), ("norm3_b", C.c_void_p)]


class isi_prior_w(C.Structure):
    _fields_ = [("d_model", C.c_int), ("nhead", C.c_int), ("dim_feedforward", C.c_int), ("n_layers", C.c_int),
                ("n_class", C.c_int), ("Cd", C.c_int), ("Ed", C.c_int), ("Ce", C.c_int), ("Ee", C.c_int),
                ("layers", isi_decoder_layer_w * ISI_MAX_LAYERS), ("logits_w", C.c_void_p),
                ("logits_b", C.c_void_p), ("embed_table", C.c_void_p), ("eff_dim", C.c_int)]


class isi_prior_state(C.Structure):
    _fields_ = [("x_seq", C.c_void_p), ("kv_cache", C.c_void_p), ("memory_kv", C.c_void_p), ("codes", C.c_void_p),
                ("mask", C.c_void_p), ("uniforms", C.c_void_p), ("scratch", C.c_void_p),
                ("scratch_floats", C.c_size_t), ("S_t", C.c_int), ("S_src", C.c_int), ("S", C.c_int),
                ("B", C.c_int), ("start_len", C.c_int)]


class isi_reduce_job(C.Structure):
    _fields_ = [("partial", C.c_void_p), ("out", C.c_void_p), ("n", C.c_int64), ("stride", C.c_int64),
                ("nsplit", C.c_int32), ("accumulate", C.c_int32), ("vec", C.c_int32),
                ("map_K", C.c_int32), ("map_Kpad", C.c_int32), ("map_cin", C.c_int32), ("map_taps", C.c_int32),
                ("map_keep", C.c_int32)]


# name -> (restype, argtypes); must list every symbol include/isi_hip.h declares
_P = C.c_void_p
SIGNATURES = {
    "isi_version": (C.c_char_p, []),
    "isi_last_error": (C.c_char_p, []),
    "isi_abi_struct_bytes": (C.c_size_t, [C.c_int]),
    "isi_decoder_tail_f32": (C.c_int, [_P, _P, _P, _P, _P, _P, C.POINTER(isi_dst), C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_int, _P]),
    "isi_knob_set": (C.c_int, [C.c_char_p, C.c_int]),
    "isi_knob_get": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "isi_relu_inplace_f32": (C.c_int, [_P, C.c_int64, _P]),
    "isi_prof_enable": (C.c_int, [C.c_int]),
    "isi_prof_num_kernels": (C.c_int, []),
    "isi_prof_kernel_name": (C.c_char_p, [C.c_int]),
    "isi_prof_read": (C.c_int, [C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_double),
                                C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "isi_pack_conv_weight_f32": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "isi_pack_conv_weight_w16_f32": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "isi_pack_conv_dgrad_weight_f32": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "isi_pack_linear_wT_bf16": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "isi_pack_linear_wT_bf16_multi": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "isi_pack_multi": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "isi_packed_conv_weight_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "isi_pack_convT_k4s2_weight_f32": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "isi_split_conv_weight_f16": (C.c_int, [_P, _P, C.c_int64, _P]),
    "isi_pair_encode_f32": (C.c_int, [_P, _P, C.c_int64, _P]),
    "isi_pair_decode_f32": (C.c_int, [_P, _P, C.c_int64, _P]),
    "isi_packed_convT_k4s2_weight_floats": (C.c_size_t, [C.c_int, C.c_int]),
    "isi_pack_codebook_f32": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P]),
    "isi_conv2d_f32": (C.c_int, [C.POINTER(isi_src), C.POINTER(isi_src), _P, _P, C.POINTER(isi_src),
                                 C.POINTER(isi_dst), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "isi_conv_transpose2d_k4s2_f32": (C.c_int, [C.POINTER(isi_src), _P, _P, C.POINTER(isi_dst), C.c_int,
                                                C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "isi_conv2d_gated_f32": (C.c_int, [C.POINTER(isi_src), C.POINTER(isi_src), _P, _P, C.POINTER(isi_src), _P,
                                       C.POINTER(isi_dst), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "isi_linear_f32": (C.c_int, [C.POINTER(isi_linear_args), _P]),
    "isi_conv_transpose2d_k4s2_gated_f32": (C.c_int, [C.POINTER(isi_src), _P, _P, _P, C.POINTER(isi_dst), C.c_int,
                                                      C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "isi_resblock_fusable": (C.c_int, [C.c_int, C.c_int]),
    "isi_resblock_f32": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, _P]),
    "isi_conv2d_twin_f32": (C.c_int, [C.POINTER(isi_src), C.POINTER(isi_src), _P, _P, C.POINTER(isi_dst), _P, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "isi_conv_transpose2d_k4s2_twin_f32": (C.c_int, [C.POINTER(isi_src), _P, _P, C.POINTER(isi_dst), _P, C.c_int, C.c_int,
                                                     C.c_int, C.c_int, C.c_int, _P]),
    "isi_resblock_tape_f32": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, _P]),
    "isi_conv2d_pair_route": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "isi_conv_wgrad_halo_route": (C.c_int, [C.c_int] * 9),
    "isi_conv_transpose2d_pair_route": (C.c_int, [C.c_int, C.c_int]),
    "isi_resblock_pair_route": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "isi_spec_polar_f32": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "isi_spec_finish_f32": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "isi_spec_inverse_prepare_f32": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "isi_spec_to_stft_f32": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int, C.c_int, _P]),
    "isi_spec_affine_mask_f32": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float,
                                           C.c_float, C.c_int, _P]),
    "isi_spec_distance_fwd_f32": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _P]),
    "isi_spec_distance_bwd_f32": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _P]),
    "isi_spec_to_stft_bwd_f32": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_int, C.c_int, _P]),
    "isi_spec_inverse_prepare_bwd_f32": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "isi_overlap_add_f32": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, _P]),
    "isi_rel_attention_f32": (C.c_int, [C.POINTER(isi_attn_args), _P]),
    "isi_rel_attention_workspace_bytes": (C.c_size_t, [C.POINTER(isi_attn_args)]),
    "isi_rel_attention_bwd_workspace_floats": (C.c_size_t, [C.POINTER(isi_attn_args)]),
    "isi_rel_attention_bwd_f32": (C.c_int, [C.POINTER(isi_attn_bwd_args), _P]),
    "isi_layernorm_bwd_workspace_floats": (C.c_size_t, [C.c_int64, C.c_int]),
    "isi_layernorm_bwd_f32": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_int64, C.c_int, C.c_float, _P]),
    "isi_label_smoothing_loss_f32": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_float,
                                               _P]),
    "isi_rel_attention_decode_f32": (C.c_int, [C.POINTER(isi_attn_args), C.c_int, _P, _P]),
    "isi_rel_attention_decode_workspace_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "isi_layernorm_f32": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_int, C.c_float, _P]),
    "isi_set_dropout_seed_base": (C.c_int, [_P]),
    "isi_layernorm_dropout_f32": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_uint64, _P]),
    "isi_layernorm_dropout_bwd_f32": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int64, C.c_int, C.c_float,
                                                C.c_float, C.c_uint64, _P]),
    "isi_linear_rows_f32": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, _P]),
    "isi_sample_row_f32": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float, _P, _P, _P,
                                     _P]),
    "isi_prior_decode_scratch_floats": (C.c_size_t, [C.POINTER(isi_prior_w), C.c_int]),
    "isi_prior_sample_run": (C.c_int, [C.POINTER(isi_prior_w), C.POINTER(isi_prior_state), C.c_int, C.c_int,
                                       C.c_float, C.c_int, C.c_float, _P]),
    "isi_conv_wgrad_workspace_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "isi_conv_wgrad_f32": (C.c_int, [C.POINTER(isi_src), C.POINTER(isi_src), _P, _P, _P, _P, C.c_size_t, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "isi_conv_wgrad_torch_f32": (C.c_int, [C.POINTER(isi_src), C.POINTER(isi_src), _P, _P, C.c_int, _P, _P,
                                           C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_int, _P]),
    "isi_conv_wgrad_deferred_f32": (C.c_int, [C.POINTER(isi_src), C.POINTER(isi_src), _P, _P, C.c_int, _P, _P,
                                              C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_int, C.c_int, _P, C.POINTER(isi_reduce_job), C.POINTER(C.c_int)]),
    "isi_reduce_jobs_f32": (C.c_int, [C.POINTER(isi_reduce_job), C.c_int, _P]),
    "isi_relu_bwd_f32": (C.c_int, [_P, _P, C.c_int64, _P]),
    "isi_axpy_f32": (C.c_int, [_P, _P, C.c_float, C.c_int64, _P]),
    "isi_vq_bwd_f32": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, _P]),
    "isi_pad_channels4_f32": (C.c_int, [C.POINTER(isi_src), _P, C.c_int, C.c_int, C.c_int, _P]),
    "isi_add_gate_rows_f32": (C.c_int, [_P, _P, C.c_int64, _P, _P, C.c_int64, C.c_int, _P]),
    "isi_vq_bwd_rows_f32": (C.c_int, [_P, _P, C.c_int64, _P, _P, _P, C.c_int64, C.c_int, _P]),
    "isi_colsum_num_partials": (C.c_int, [C.c_int64]),
    "isi_colsum_f32": (C.c_int, [_P, C.c_int64, _P, _P, C.c_int64, C.c_int, _P]),
    "isi_vq_embed_sum_workspace_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int64]),
    "isi_vq_embed_sum_f32": (C.c_int, [_P, _P, _P, _P, C.c_size_t, C.c_int64, C.c_int, C.c_int, _P]),
    "isi_vq_ema_update_f32": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_float, C.c_float, _P]),
    "isi_vq_nearest_f32": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int64, C.c_int, C.c_int, _P]),
    "isi_vq_nearest_flags_f32": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int64, C.c_int, C.c_int, C.c_int, _P]),
    "isi_vq_conv1x1_nearest_f32": (C.c_int, [C.POINTER(isi_src), C.POINTER(isi_src), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                             C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "isi_vq_conv1x1_nearest_tape_f32": (C.c_int, [C.POINTER(isi_src), C.POINTER(isi_src), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                                  _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "isi_vq_conv1x1_workspace_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "isi_vq_conv1x1_fusable": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "isi_vq_pack_fragments_f32": (C.c_int, [_P, _P, C.c_int, _P]),
    "isi_decode_stage_workspace_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "isi_decode_stage_f32": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, C.c_int, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_float, _P, C.c_size_t, _P]),
    "isi_mse_loss_num_partials": (C.c_int, [C.c_int64]),
    "isi_mse_loss_f32": (C.c_int, [_P, _P, C.c_int64, _P, _P, _P]),
    "isi_mse_loss_bwd_f32": (C.c_int, [_P, _P, _P, C.c_int64, _P, _P, _P]),
    "isi_vq_num_partials": (C.c_int, [C.c_int64]),
    "isi_vq_finalize_f32": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int64, C.c_int, _P, _P]),
    "isi_embed_code_f32": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int, C.c_int, _P]),
    "isi_vqvae_workspace_bytes": (C.c_size_t, [C.POINTER(isi_vqvae_w), C.c_int, C.c_int, C.c_int]),
    "isi_vqvae_pair_activations": (C.c_int, [C.POINTER(isi_vqvae_w)]),
    "isi_vqvae_run": (C.c_int, [C.POINTER(isi_vqvae_w), C.c_int, _P, C.c_int, C.c_int, C.c_int,
                                C.POINTER(isi_vqvae_out), _P, C.c_size_t, _P]),
}

_lib = None
_lock = threading.Lock()


def library_path() -> pathlib.Path:
    return pathlib.Path(os.environ.get("ISI_HIP_LIBRARY", str(_LIB_PATH)))


def lib() -> C.CDLL:
    """Load libisi_hip.so once; raises HipLibraryError if it is not built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                path = library_path()
                if not path.exists():
                    raise HipLibraryError(
                        f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "or `make -C interactive-spectrogram-inpainting_amd/csrc`; there is no fallback path")
                handle = C.CDLL(str(path))
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(handle, name)
                    fn.restype = res
                    fn.argtypes = args
                structs = [isi_src, isi_dst, isi_conv_w, isi_encoder_w, isi_decoder_w, isi_codebook_w,
                           isi_vqvae_w, isi_vqvae_out, isi_attn_args, isi_prior_w, isi_prior_state, isi_attn_bwd_args,
                           isi_reduce_job]
                for i, st in enumerate(structs):
                    if handle.isi_abi_struct_bytes(i) != C.sizeof(st):
                        raise HipLibraryError(f"ABI mismatch for {st.__name__}: library "
                                              f"{handle.isi_abi_struct_bytes(i)} B, binding {C.sizeof(st)} B")
                _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().isi_last_error().decode("utf-8", "replace")
        raise HipLibraryError(f"{what} failed with code {rc}: {msg}")


# ---- weight versions.  Packed weights (GEMM operands, split-f16 copies, transposes for the input gradients) are cached
# per weight VERSION.  Every in-place torch op bumps `tensor._version` -- but torch's fused optimizers
# (`Adam(fused=True)`) update the parameters without touching it: the caches would serve the weights of the first
# step for ever.  A global optimizer-step hook therefore counts steps, and the count is part of every cache key: any
# optimizer step invalidates every packed weight (at worst a re-pack that was not needed).
_OPTIMIZER_STEPS = 0


def _on_optimizer_step(*_args, **_kwargs) -> None:
    global _OPTIMIZER_STEPS
    _OPTIMIZER_STEPS += 1


try:
    from torch.optim.optimizer import register_optimizer_step_post_hook as _register_step_hook
    _register_step_hook(_on_optimizer_step)
except ImportError:   # a torch without global optimizer hooks: version counters only (do not use fused optimizers there)
    pass


def optimizer_steps() -> int:
    return _OPTIMIZER_STEPS


def version_of(t) -> tuple:
    """Cache-key component for anything derived from tensor `t`'s values."""
    return (t._version, _OPTIMIZER_STEPS)


class knob:
    """`with knob("ISI_CONV_FLUSH", 0): ...` -- set an execution switch of the library (isi_knob_set) for the block and
    restore it afterwards.  The switches select between kernels that compute the same result; the library reads the
    environment variables of the same names only once, at first use."""

    def __init__(self, name: str, value: int):
        self.name, self.value = name.encode(), int(value)

    def __enter__(self):
        old = C.c_int(0)
        check(lib().isi_knob_get(self.name, C.byref(old)), "isi_knob_get")
        self.old = old.value
        check(lib().isi_knob_set(self.name, self.value), "isi_knob_set")
        return self

    def __exit__(self, *exc):
        check(lib().isi_knob_set(self.name, self.old), "isi_knob_set")
        return False


def stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def require_gpu(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise HipLibraryError(
            f"{name} lives on {t.device}: this package computes only on an MI355X "
            "(HIP kernels, no CPU fallback)")
    if t.dtype != torch.float32 and t.dtype != torch.int64:
        raise HipLibraryError(f"{name} has dtype {t.dtype}; expected float32 / int64")


def src_nhwc(t: torch.Tensor) -> isi_src:
    """Descriptor of a dense channels-last [B,H,W,C] tensor."""
    B, H, W, Cc = t.shape
    return isi_src(t.data_ptr(), Cc, H * W * Cc, 1, W * Cc, Cc)


def src_nchw_view(t: torch.Tensor) -> isi_src:
    """Descriptor of any 4-D tensor indexed as [B,C,H,W] (arbitrary strides)."""
    sn, sc, sh, sw = t.stride()
    return isi_src(t.data_ptr(), t.shape[1], sn, sc, sh, sw)


def dst_nchw_view(t: torch.Tensor) -> isi_dst:
    sn, sc, sh, sw = t.stride()
    return isi_dst(t.data_ptr(), sn, sc, sh, sw)
