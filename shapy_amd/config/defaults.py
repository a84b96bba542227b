"""Default configuration tree of the SHAPY regressor hot path.

Mirrors the *keys and default values* of the reference's structured config
(``regressor/human_shape/config/defaults.py:52-112``, ``network_defaults.py``,
``body_model.py``, ``datasets_defaults.py:19-50``) for everything the model
constructors and the demo/evaluate entry points read.
Training-only subtrees (losses, optimiser) are carried as open dicts: YAML files
may set anything below them, nothing on the inference path reads them.
"""
from .node import ConfigNode


def _pose(type_='cont-rot-repr'):
    return {'create': True, 'requires_grad': True, 'type': type_}


def _pose_pca():
    d = _pose()
    d['pca'] = {'num_comps': 12, 'flat_hand_mean': False}
    return d


def _activation():
    return {'type': 'relu', 'inplace': True,
            'leaky_relu': {'negative_slope': 0.01},
            'prelu': {'num_parameters': 1, 'init': 0.25},
            'elu': {'alpha': 1.0}}


def _normalization():
    return {'type': 'batch-norm',
            'batch_norm': {'eps': 1e-05, 'momentum': 0.1, 'affine': True,
                           'track_running_stats': True},
            'group_norm': {'num_groups': 32, 'eps': 1e-05, 'affine': True}}


def _stage(num_modules=1, num_branches=1, num_blocks=(4,), num_channels=(64,),
           block='BOTTLENECK'):
    return {'num_modules': num_modules, 'num_branches': num_branches,
            'num_blocks': list(num_blocks), 'num_channels': list(num_channels),
            'block': block, 'fuse_method': 'SUM'}


def _hrnet():
    # network_defaults.py:92-132
    return {
        'use_old_impl': False,
        'pretrained_layers': ['*'],
        'pretrained_path': '../data/hrnet_v2/hrnetv2_w48_imagenet_pretrained.pth',
        'stage1': _stage(),
        'stage2': _stage(1, 2, (4, 4), (48, 96), 'BASIC'),
        'stage3': _stage(4, 3, (4, 4, 4), (48, 96, 192), 'BASIC'),
        'stage4': _stage(3, 4, (4, 4, 4, 4), (48, 96, 192, 384), 'BASIC'),
    }


def _backbone(type_='resnet50'):
    return {'type': type_, 'pretrained': True,
            'resnet': {'replace_stride_with_dilation': [False, False, False]},
            'hrnet': _hrnet()}


def _camera():
    return {'type': 'weak-persp', 'pos_func': 'softplus',
            'weak_persp': {'regress_scale': True, 'regress_translation': True,
                           'mean_scale': 0.9, 'scale_first': False},
            'perspective': {'regress_translation': False, 'regress_rotation': False,
                            'regress_focal_length': False, 'focal_length': 5000.0}}


def _mlp():
    return {'layers': [1024, 1024], 'activation': _activation(),
            'normalization': _normalization(), 'preactivated': False,
            'dropout': 0.0, 'init_type': 'xavier', 'gain': 0.01, 'bias_init': 0.0}


def _hmr_like():
    return {'type': 'mlp', 'feature_key': 'avg_pooling', 'append_params': True,
            'num_stages': 3, 'pose_last_stage': True, 'detach_mean': False,
            'learn_mean': False, 'backbone': _backbone(), 'camera': _camera(),
            'mlp': _mlp()}


def _smpl_head(groups):
    d = _hmr_like()
    d.update({
        'compute_measurements': True, 'meas_definition_path': '',
        'meas_vertices_path': '',
        'use_b2a': True, 'b2a_males_checkpoint': '', 'b2a_females_checkpoint': '',
        'use_a2b': True, 'num_attributes': 15,
        'a2b_males_checkpoint': '', 'a2b_females_checkpoint': '',
        'groups': [list(groups)], 'joints_to_exclude': [],
    })
    return d


def _smplh_head(groups):
    d = _smpl_head(groups)
    d['predict_hands'] = True
    return d


def _smplx_head():
    d = _smplh_head(('betas', 'expression', 'global_rot', 'body_pose',
                     'left_hand_pose', 'right_hand_pose', 'jaw_pose', 'camera'))
    d['predict_face'] = True
    return d


def _network():
    return {
        'type': 'expose', 'use_sync_bn': True,
        'hmr': _hmr_like(),
        'smpl': _smpl_head(('betas', 'global_rot', 'body_pose', 'camera')),
        'smplh': _smplh_head(('betas', 'global_rot', 'body_pose', 'left_hand_pose',
                              'right_hand_pose', 'camera')),
        'smplx': _smplx_head(),
        'expose': _smplx_head(),
    }


def _abstract_body_model():
    return {'extra_joint_path': '', 'v_template_path': '', 'mean_pose_path': '',
            'shape_mean_path': '', 'use_compressed': True,
            'learn_joint_regressor': False}


def _smpl_model():
    d = _abstract_body_model()
    d.update({'ext': 'pkl', 'use_feet_keypoints': True, 'use_face_keypoints': True,
              'j14_regressor_path': '',
              'betas': {'create': True, 'requires_grad': True, 'num': 10},
              'global_rot': _pose(), 'body_pose': _pose(),
              'translation': {'create': True, 'requires_grad': True},
              'head_verts_ids_path': ''})
    return d


def _smplh_model():
    d = _smpl_model()
    d.update({'left_hand_pose': _pose_pca(), 'right_hand_pose': _pose_pca()})
    return d


def _smplx_model():
    d = _smplh_model()
    d.update({'ext': 'npz', 'use_face_contour': False,
              'expression': {'create': True, 'requires_grad': True, 'num': 10},
              'jaw_pose': _pose(), 'leye_pose': _pose(), 'reye_pose': _pose(),
              'hand_vertex_ids_path': ''})
    return d


def _body_model():
    return {'type': 'smplx', 'model_folder': 'models', 'smpl': _smpl_model(),
            'smplh': _smplh_model(), 'smplx': _smplx_model()}


def _transforms():
    # datasets_defaults.py:19-50 (crop_size 256: SURVEY.md F3)
    return {'flip_prob': 0.0, 'max_size': 1080, 'crop_size': 256,
            'scale_factor_min': 1.0, 'scale_factor_max': 1.0, 'scale_factor': 0.0,
            'scale_dist': 'uniform', 'noise_scale': 0.0, 'rotation_factor': 0.0,
            'mean': [0.485, 0.456, 0.406], 'std': [0.229, 0.224, 0.225]}


def _dataset_part():
    return {'splits': {'train': [], 'val': [], 'test': []},
            'num_workers': {'train': 8, 'val': 2, 'test': 2},
            'transforms': _transforms(),
            'sampler': {'ratio_2d': 0.5, 'use_equal_sampling': True,
                        'importance_key': 'weight', 'balance_genders': True},
            'openpose': {'data_folder': 'data/openpose', 'img_folder': 'images',
                         'keyp_folder': 'keypoints', 'keyp_format': 'openpose25_v1',
                         'binarization': True, 'body_thresh': 0.05,
                         'hand_thresh': 0.2, 'head_thresh': 0.3,
                         'use_face_contour': True, 'metrics': ['mpjpe14']}}


def default_config():
    return ConfigNode({
        'num_gpus': 1, 'local_rank': 0, 'use_cuda': True, 'is_training': True,
        'logger_level': 'info', 'use_half_precision': False,
        'output_folder': 'output', 'summary_folder': 'summaries',
        'results_folder': 'results', 'code_folder': 'code',
        'summary_steps': 100, 'img_summary_steps': 100,
        'hd_img_summary_steps': 1000, 'imgs_per_row': 2, 'backend': 'nccl',
        'part_key': 'pose', 'degrees': [90, 180, 270],
        'j14_regressor_path': '', 'pretrained': '', 'use_adv_training': False,
        'checkpoint_folder': 'checkpoints', 'checkpoint_steps': 1000,
        'eval_steps': 500, 'float_dtype': 'float32',
        'max_duration': float('inf'), 'max_iters': float('inf'),
        'body_vertex_ids_path': '',
        'network': _network(),
        'optim': {},
        'body_model': _body_model(),
        'datasets': {'batch_size': 64, 'pose_shape_ratio': 0.5,
                     'use_equal_sampling': True, 'use_packed': False,
                     'pose': _dataset_part(), 'shape': _dataset_part()},
        'losses': {'body': {}},
        'evaluation': {'body': {
            'v2v': ['procrustes', 'scale', 'translation'],
            'v2v_t': ['scale', 'translation'],
            'mpjpe': {'alignments': ['root', 'procrustes'],
                      'root_joints': ['left_hip', 'right_hip']},
            'fscores_thresh': [0.01, 0.02, 0.05, 0.075, 0.1],
            'p2p_t': {'input_point_regressor_path': '',
                      'target_point_regressor_path': '', 'align': True}}},
        'run_final_evaluation_on_validation_set': False,
    })


conf = default_config()
