// External names of the fp16 instantiation of gemm.hip / attention.hip / rowops.hip / encoder_api.hip (common.h, TT_F16).
#pragma once
// launch functions (encoder.h)
#define tt_gemm_launch tt_gemm_launch_f16
#define tt_scan_gemm_launch tt_scan_gemm_launch_f16
#define tt_scan_gemm_sample_launch tt_scan_gemm_sample_launch_f16
#define tt_gemm_skinny_enabled tt_gemm_skinny_enabled_f16
#define tt_attention_launch tt_attention_launch_f16
#define tt_attention_cls_launch tt_attention_cls_launch_f16
#define tt_absmax_launch tt_absmax_launch_f16
#define tt_embed_ln_launch tt_embed_ln_launch_f16
#define tt_layernorm_launch tt_layernorm_launch_f16
#define tt_gather_rows_launch tt_gather_rows_launch_f16
#define tt_quantize_rows_launch tt_quantize_rows_launch_f16
#define tt_adjacent_cosine_launch tt_adjacent_cosine_launch_f16
#define tt_cls_pool_l2norm_launch tt_cls_pool_l2norm_launch_f16
#define tt_head_out_sigmoid_launch tt_head_out_sigmoid_launch_f16
#define tt_mean_pool_l2norm_launch tt_mean_pool_l2norm_launch_f16
// C ABI (include/tt_hip.h declares the _f16 names)
#define tt_encoder_workspace_bytes tt_encoder_workspace_bytes_f16
#define tt_encoder_cls_workspace_bytes tt_encoder_cls_workspace_bytes_f16
#define tt_encoder_forward tt_encoder_forward_f16
#define tt_encoder_forward_cls tt_encoder_forward_cls_f16
#define tt_embed_pool tt_embed_pool_f16
#define tt_embed_pool_mean tt_embed_pool_mean_f16
#define tt_rerank_head tt_rerank_head_f16
#define tt_gemm_bf16 tt_gemm_f16
#define tt_layernorm_bf16 tt_layernorm_f16
#define tt_attention_varlen tt_attention_varlen_f16
// x3_path.hip a second time: the split planes are fp16 there ("f16x3": 22 significand bits per operand instead of bf16x3's 16)
#define tt_encoder_x3_workspace_bytes tt_encoder_x3_workspace_bytes_f16
#define tt_encoder_x3_cls_workspace_bytes tt_encoder_x3_cls_workspace_bytes_f16
#define tt_encoder_forward_x3 tt_encoder_forward_x3_f16
#define tt_encoder_forward_x3_cls tt_encoder_forward_x3_cls_f16
#define tt_rerank_head_x3 tt_rerank_head_x3_f16
#define tt_split_planes tt_split_planes_f16
#define tt_gemm_x3 tt_gemm_x3_f16
#define tt_attention_x3 tt_attention_x3_f16
