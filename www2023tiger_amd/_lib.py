"""ctypes binding of libtiger_hip.so (include/tiger_hip.h).

There is no CPU fallback: if the shared library is missing or a symbol is absent
the import of this module raises, and every op of the package fails with it.
"""
import ctypes as C
import os

# torch first: its wheel bundles the HIP runtime (libamdhip64) that owns the tensors and streams
# handed to the library; loading libtiger_hip.so before it would bind a second runtime copy.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('TIGER_HIP_LIB') or os.path.join(_HERE, 'csrc', 'libtiger_hip.so')  # override: experiments only

TG_OK, TG_EINVAL, TG_EUNSUPPORTED, TG_EWORKSPACE, TG_EHIP = 0, -1, -2, -3, -4
ERR_PAST_MEMORY, ERR_DUPLICATE_IDS, ERR_UNUSED_MESSAGE = 1, 2, 4
ERR_MSG_BEFORE_MEM, ERR_MSG_TS_MISMATCH, ERR_EVENT_BEFORE_MEM = 8, 16, 32

# the reference's exception text for each device-side invariant bit
ERR_TEXT = {
    ERR_PAST_MEMORY: 'You are not allowed to modify past memory.',
    ERR_DUPLICATE_IDS: 'Duplicate node ids are not allowed.',
    ERR_UNUSED_MESSAGE: 'Node has unused messages.',
    ERR_MSG_BEFORE_MEM: 'Messages happened later than memory updating.',
    ERR_MSG_TS_MISMATCH: "Messages' ts should be equal to last update ts when using left memory as msg source.",
    ERR_EVENT_BEFORE_MEM: 'Events occur before the udpated memory.',
}

vp, i32, i64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t


class TgTcsr(C.Structure):
    _fields_ = [('num_node', i64), ('num_entry', i64), ('indptr', vp), ('ts', vp), ('nbr', vp), ('eid', vp)]


class TgLinear(C.Structure):
    _fields_ = [('w', vp), ('b', vp)]


class TgModel(C.Structure):
    _fields_ = [
        ('n_nodes', i64), ('d', i32), ('d_e', i32), ('n_neighbors', i32), ('n_head', i32),
        ('msg_src', i32), ('upd_src', i32), ('tsfm', i32), ('upd_fn', i32),
        ('left_vals', vp), ('left_ts', vp), ('left_active', vp),
        ('right_vals', vp), ('right_ts', vp), ('right_active', vp),
        ('msg_vals', vp), ('msg_ts', vp), ('has_msg', vp),
        ('nfeats', vp), ('efeats', vp), ('te_freq', vp), ('te_phase', vp),
        ('tsfm1', TgLinear), ('tsfm2', TgLinear),
        ('gru_w_ih', vp), ('gru_w_hh', vp), ('gru_b_ih', vp), ('gru_b_hh', vp),
        ('upd_fc1', TgLinear), ('upd_fc2', TgLinear),
        ('attn_wq', vp), ('attn_wk', vp), ('attn_wv', vp), ('attn_b_in', vp),
        ('attn_out', TgLinear), ('attn_fc1', TgLinear), ('attn_fc2', TgLinear), ('attn_fused', vp),
        ('pending_vals', vp), ('row_of', vp), ('g_table', vp), ('c_table', vp),
    ]


class TgSeqRestarter(C.Structure):
    _fields_ = [
        ('hist_len', i32), ('n_head', i32), ('te_freq', vp), ('te_phase', vp), ('anony_emb', vp),
        ('in_proj_w', vp), ('in_proj_b', vp), ('out_proj', TgLinear), ('out_fn', TgLinear),
        ('fc1', TgLinear), ('fc2', TgLinear), ('nfeats_zero', i32), ('reserved', i32), ('ta_cached', vp),
    ]


class TgStepIo(C.Structure):
    _fields_ = [
        ('B', i64), ('src', vp), ('dst', vp), ('neg', vp), ('ts', vp), ('eids', vp),
        ('h', vp), ('l1_nids', vp), ('l1_eids', vp), ('l1_ts', vp), ('involved', vp), ('counts', vp),
        ('h_prev_left', vp), ('h_prev_right', vp), ('err', vp),
        ('offset_dev', vp), ('advance', i32), ('embed_only', i32), ('profiler', vp), ('h_new', vp),
        ('ws_is_clean', i32), ('rows_hint', i32), ('lazy', vp), ('collate_only', i32), ('eager_copy', i32), ('lean', i32), ('strategy', i32),
        ('inner', vp), ('stream_len', i64), ('prefetch_state', vp),
        ('dbg_l1_nids', vp), ('dbg_l1_eids', vp), ('dbg_l1_ts', vp), ('mt_state', vp),
    ]


class TgLazyRestart(C.Structure):
    _fields_ = [('static_left', vp), ('static_right', vp), ('trigger', vp), ('n_trigger', i64), ('batch_dev', vp),
                ('restarting_dev', vp), ('uptodate', vp), ('list', vp), ('tmin', vp), ('keep_msg_bits', i32),
                ('reserved', i32)]


class TgRestartRun(C.Structure):
    """tiger_hip.h: tg_restart_run - the restart-mode evaluation pass over consecutive batches as one call"""
    _fields_ = [('group', i32), ('reserved', i32), ('pass_io', vp * 16), ('pass_ws', vp * 16), ('pass_ws_bytes', sz * 16),
                ('g_restart', vp), ('offsets', vp), ('batch_dev', vp), ('count_host', vp * 16), ('cap', i64), ('rows_cap', i64),
                ('ids', vp * 2), ('h_left', vp * 2), ('h_right', vp * 2), ('prev_ts', vp * 2),
                ('fwd_nodes', i64), ('fwd_ws', vp), ('fwd_ws_bytes', sz), ('static_left', vp), ('static_right', vp),
                ('gtab_ws', vp), ('gtab_ws_bytes', sz),
                ('pos_scores', vp), ('neg_scores', vp), ('n_restarted', vp), ('stream_len', i64), ('first_offset', i64)]


TG_MAX_RANKS = 16


class TgPart(C.Structure):
    """tiger_hip.h: tg_part - one rank's view of the partitioned multi-GPU step (windows, plan tables, staging)"""
    _fields_ = [
        ('world', i32), ('rank', i32), ('n_steps', i64), ('Bg', i64), ('step_dev', vp), ('cur_step', vp),
        ('pull_in', vp * TG_MAX_RANKS), ('push_in', vp * TG_MAX_RANKS), ('flags', vp * TG_MAX_RANKS),
        ('pull_max', i64), ('push_max', i64), ('ticket', vp), ('err', vp),
        ('g_src', vp), ('g_dst', vp), ('g_eids', vp), ('ts32', vp), ('left_row', vp),
        ('serve_cap', i64), ('n_serve', vp), ('serve_row', vp), ('serve_kind', vp), ('serve_peer', vp), ('serve_slot', vp),
        ('adopt_row', vp), ('adopt_kind', vp),
        ('req_cap', i64), ('n_req', vp), ('req_node', vp), ('req_row', vp), ('n_unmap', vp), ('unmap_node', vp),
        ('push_cap', i64), ('n_push', vp), ('push_src', vp), ('push_peer', vp), ('push_slot', vp),
        ('mine_cap', i64), ('n_mine', vp), ('mine_node', vp), ('mine_index', vp), ('mine_row', vp),
        ('st_src', vp), ('st_dst', vp), ('st_eids', vp), ('st_left_row', vp), ('st_mine_node', vp), ('st_mine_index', vp),
        ('st_mine_row', vp), ('st_ts32', vp), ('st_mine32', vp), ('st_n_mine', vp),
        ('owner', vp), ('row_of', vp),
    ]


class TgWritebackIo(C.Structure):
    _fields_ = [
        ('Bg', i64), ('src', vp), ('dst', vp), ('ts', vp), ('eids', vp), ('offset_dev', vp), ('advance', i32),
        ('reserved', i32), ('rows', vp), ('left_row', vp), ('new_row', vp), ('err', vp),
        ('owner', vp), ('my_rank', i32), ('new_from_pending', i32),
        ('upos', vp), ('index', vp), ('n_upos_dev', vp), ('ts32', vp),
    ]


class TgScoreParams(C.Structure):
    _fields_ = [('hit_type', i32), ('n_hit_rows', i32), ('hit_emb', vp), ('fc1', TgLinear), ('fc2', TgLinear)]


class TgTrainIo(C.Structure):
    _fields_ = [
        ('step', TgStepIo), ('score', vp), ('grads', vp), ('score_grads', vp), ('losses', vp),
        ('pos_scores', vp), ('neg_scores', vp), ('flags', vp), ('restarter', i32), ('reserved', i32),
        ('seq', vp), ('seq_grads', vp), ('static_left', vp), ('static_right', vp),
        ('static_left_grad', vp), ('static_right_grad', vp),
        ('dropout_p', C.c_float), ('reserved2', i32), ('rng', vp), ('inner_grads', vp),
    ]


class TgAdamSeg(C.Structure):
    _fields_ = [('p', vp), ('g', vp), ('m', vp), ('v', vp), ('n', i64), ('group', i32), ('grad_scale', C.c_float)]


P = C.POINTER
# name -> (restype, argtypes); every symbol include/tiger_hip.h declares
SIGNATURES = {
    'tg_abi_version': (C.c_int, []),
    'tg_last_hip_error': (C.c_char_p, []),
    'tg_tcsr_build_host': (C.c_int, [i64, vp, vp, vp, vp, i64, vp, vp, vp, vp]),
    'tg_rand_edge_pairs_host': (C.c_int, [vp, i64, i64, i64, vp, vp]),
    'tg_rand_edge_pairs': (C.c_int, [vp, i64, i64, i64, vp, vp, vp, vp, vp]),
    'tg_tcsr_build_device_workspace_bytes': (sz, [i64, i64]),
    'tg_tcsr_build_device': (C.c_int, [i64, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, sz, vp]),
    'tg_sample_recent_edges': (C.c_int, [P(TgTcsr), i64, vp, vp, i32, vp, vp, vp, vp, vp, vp]),
    'tg_sample_recent_nodes': (C.c_int, [P(TgTcsr), i64, vp, vp, i32, vp, vp, vp, vp, vp]),
    'tg_sample_uniform': (C.c_int, [P(TgTcsr), i64, vp, vp, i32, vp, vp, vp, vp, vp, vp]),
    'tg_hits': (C.c_int, [i64, i32, vp, vp, vp, vp]),
    'tg_anonymized_reindex': (C.c_int, [i64, i32, vp, vp, vp]),
    'tg_bitmap_words': (i64, [i64]),
    'tg_bitmap_mark': (C.c_int, [i64, vp, vp, i64, vp]),
    'tg_unique_compact_workspace_bytes': (sz, [i64]),
    'tg_unique_compact': (C.c_int, [vp, vp, i64, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, sz, vp]),
    'tg_flag_bytes': (i64, [i64]),
    'tg_flags_mark': (C.c_int, [i64, vp, vp, i64, vp]),
    'tg_select_latest_workspace_bytes': (sz, [i64, i64]),
    'tg_select_latest': (C.c_int, [i64, vp, vp, i32, i64, vp, vp, vp, vp, sz, vp]),
    'tg_time_encode': (C.c_int, [i64, vp, i32, vp, vp, vp, vp]),
    'tg_gather_rows': (C.c_int, [i64, vp, i32, vp, vp, vp, vp, vp]),
    'tg_memory_scatter': (C.c_int, [i64, vp, vp, vp, i32, vp, vp, vp, vp, vp, i32, vp, vp]),
    'tg_memory_scatter2': (C.c_int, [i64, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, i32, vp, vp]),
    'tg_consume_update_right_rows': (C.c_int, [P(TgModel), vp, vp, i64, vp, vp, vp, vp]),
    'tg_linear_fwd': (C.c_int, [i64, vp, i32, P(TgLinear), i32, i32, vp, vp]),
    'tg_linear_bwd_workspace_bytes': (sz, [i32, i32]),
    'tg_linear_bwd': (C.c_int, [i64, vp, i32, vp, i32, vp, vp, vp, vp, vp, sz, vp]),
    'tg_gru_fwd': (C.c_int, [i64, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp]),
    'tg_mailbox_consume_gather': (C.c_int, [P(TgModel), vp, vp, i64, vp, vp]),
    'tg_apply_messages_workspace_bytes': (sz, [P(TgModel), i64]),
    'tg_apply_messages': (C.c_int, [P(TgModel), vp, vp, vp, i64, vp, vp, vp, sz, vp]),
    'tg_temporal_attn_workspace_bytes': (sz, [P(TgModel), i64]),
    'tg_temporal_attn_fwd': (C.c_int, [P(TgModel), i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    'tg_temporal_attn_fwd_keys': (C.c_int, [P(TgModel), i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    'tg_consume_update_right': (C.c_int, [P(TgModel), vp, vp, i64, vp, vp, vp, vp, vp]),
    'tg_gather_eff_rows': (C.c_int, [P(TgModel), i64, vp, vp, vp, vp]),
    'tg_serve_rows': (C.c_int, [P(TgModel), i64, vp, vp, i64, vp, vp, vp, vp]),
    'tg_adopt_rows': (C.c_int, [P(TgModel), i64, vp, vp, i64, vp, vp, vp, vp]),
    'tg_store_events': (C.c_int, [P(TgModel), i64, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    'tg_restart_seq_workspace_bytes': (sz, [P(TgModel), P(TgSeqRestarter), i64]),
    'tg_restart_seq_fwd': (C.c_int, [P(TgModel), P(TgSeqRestarter), i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    'tg_restart_seq_fwd_train': (C.c_int, [P(TgModel), P(TgSeqRestarter), i64, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                           C.c_float, vp, vp, sz, vp]),
    'tg_restart_apply': (C.c_int, [P(TgModel), i64, vp, vp, vp, vp, vp]),
    'tg_restart_seq_list_workspace_bytes': (sz, [P(TgModel), P(TgSeqRestarter), i64]),
    'tg_restart_seq_list': (C.c_int, [P(TgModel), P(TgTcsr), P(TgSeqRestarter), i64, vp, vp, vp, sz, vp]),
    'tg_restart_seq_list_train': (C.c_int, [P(TgModel), P(TgTcsr), P(TgSeqRestarter), i64, vp, vp, C.c_float, vp, vp, sz, vp]),
    'tg_restart_seq_list_dev': (C.c_int, [P(TgModel), P(TgTcsr), P(TgSeqRestarter), i64, vp, vp, vp, vp, sz, vp]),
    'tg_eval_restart_run': (C.c_int, [P(TgModel), P(TgTcsr), vp, vp, vp, sz, P(TgRestartRun), i64, vp]),
    'tg_restart_seq_lists_fwd': (C.c_int, [P(TgModel), P(TgTcsr), P(TgSeqRestarter), i32, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    'tg_restart_static_lists_fwd': (C.c_int, [P(TgModel), P(TgTcsr), vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
    'tg_restart_seq_list_fwd': (C.c_int, [P(TgModel), P(TgTcsr), P(TgSeqRestarter), i64, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    'tg_profiler_create': (vp, []),
    'tg_profiler_destroy': (None, [vp]),
    'tg_profiler_num_stages': (C.c_int, []),
    'tg_profiler_stage_name': (C.c_char_p, [C.c_int]),
    'tg_profiler_read': (C.c_int, [vp, vp]),
    'tg_profiler_num_kernel_slots': (C.c_int, []),
    'tg_profiler_kernel_slot_name': (C.c_char_p, [C.c_int]),
    'tg_profiler_kernel_ms': (C.c_int, [vp, P(C.c_float), P(C.c_char_p)]),
    'tg_stream_step_workspace_bytes': (sz, [P(TgModel), i64]),
    'tg_stream_step_workspace_bytes2': (sz, [P(TgModel), i64, i32]),
    'tg_stream_step_zero_bytes': (sz, [P(TgModel), i64]),
    'tg_stream_step_zero_bytes2': (sz, [P(TgModel), i64, i32]),
    'tg_stream_step_form': (i32, [P(TgModel), P(TgStepIo)]),
    'tg_part_step': (C.c_int, [P(TgModel), P(TgTcsr), P(TgStepIo), P(TgPart), vp, sz, vp, sz, vp]),
    'tg_xchg_selftest': (C.c_int, [P(TgPart), i32, i32, vp, vp]),
    'tg_xchg_alloc': (C.c_int, [sz, P(vp)]),
    'tg_xchg_clear': (C.c_int, [vp, sz]),
    'tg_xchg_free': (C.c_int, [vp]),
    'tg_ipc_export': (C.c_int, [vp, vp]),
    'tg_ipc_import': (C.c_int, [vp, P(vp)]),
    'tg_ipc_close': (C.c_int, [vp]),
    'tg_stream_step': (C.c_int, [P(TgModel), P(TgTcsr), P(TgStepIo), vp, sz, vp]),
    'tg_train_step_workspace_bytes': (sz, [P(TgModel), P(TgScoreParams), i32, vp, i64]),
    'tg_train_step_workspace_bytes2': (sz, [P(TgModel), P(TgScoreParams), i32, vp, i64, i32]),
    'tg_train_step': (C.c_int, [P(TgModel), P(TgTcsr), P(TgTrainIo), vp, sz, vp]),
    'tg_adam_step': (C.c_int, [vp, i32, i32, vp, vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, vp]),
    'tg_attn_fused_floats': (sz, [P(TgModel)]),
    'tg_attn_fuse_workspace_bytes': (sz, [P(TgModel)]),
    'tg_attn_fuse': (C.c_int, [P(TgModel), vp, vp, sz, vp]),
    'tg_attn_tile_applies': (C.c_int, [P(TgModel)]),
    'tg_attn_gtab_rows': (C.c_int, [P(TgModel), i64, vp, vp, vp, sz, vp]),
    'tg_ap_auc': (C.c_int, [i64, i32, vp, vp, vp, vp, vp, vp]),
    'tg_stream_writeback_workspace_bytes': (sz, [P(TgModel), i64]),
    'tg_stream_writeback': (C.c_int, [P(TgModel), P(TgWritebackIo), vp, sz, vp]),
}


class TigerHipError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise TigerHipError(
            f'{LIB_PATH} is missing: build it with `python __graft_entry__.py` (or `make -C www2023tiger_amd/csrc`). '
            'There is no CPU fallback for the HIP path.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export the symbol
        fn.restype = res
        fn.argtypes = args
    if lib.tg_abi_version() != 9:
        raise TigerHipError('libtiger_hip.so ABI version mismatch')
    return lib


lib = _load()


def check(rc, what=''):
    if rc == TG_OK:
        return
    names = {TG_EINVAL: 'invalid argument', TG_EUNSUPPORTED: 'unsupported configuration',
             TG_EWORKSPACE: 'workspace too small', TG_EHIP: 'HIP error'}
    msg = names.get(rc, f'error {rc}')
    if rc == TG_EHIP:
        msg += ': ' + lib.tg_last_hip_error().decode()
    raise TigerHipError(f'{what}: {msg}')


def ptr(t):
    """device (or host) pointer of a torch tensor / numpy array, None -> NULL"""
    if t is None:
        return None
    if hasattr(t, 'data_ptr'):
        return t.data_ptr()
    return t.ctypes.data


def raise_invariants(word: int):
    """Turn the device-side invariant word into the reference's ValueError."""
    for bit, text in ERR_TEXT.items():
        if word & bit:
            raise ValueError(text)
