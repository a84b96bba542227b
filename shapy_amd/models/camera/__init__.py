from .camera_projection import build_cam_proj, DEFAULT_FOCAL_LENGTH, CameraParams, WeakPerspectiveCamera
