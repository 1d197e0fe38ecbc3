import sys; sys.path.insert(0, ".")
import numpy as np
from differt2d_amd.scene import Scene
from differt2d_amd.utils import received_power
from differt2d_amd.geometry import MinPath, FermatPath
from differt2d_amd.optimize import adam
from differt2d_amd.random import PRNGKey
scene = Scene.square_scene_with_wall()
X, Y = scene.grid(n=60)
P = scene.accumulate_on_receivers_grid_over_paths(X, Y, fun=received_power, reduce_all=True, approx=True, path_cls=MinPath,
                                                  path_cls_kwargs=dict(steps=100, optimizer=adam(0.05)), key=PRNGKey(1234))
Q = scene.accumulate_on_receivers_grid_over_paths(X, Y, fun=received_power, reduce_all=True, approx=True, path_cls=MinPath,
                                                  path_cls_kwargs=dict(steps=100, optimizer=adam(0.05)), key=1234)
R_ = scene.accumulate_on_receivers_grid_over_paths(X, Y, fun=received_power, reduce_all=True, approx=True, path_cls=MinPath,
                                                  path_cls_kwargs=dict(steps=100, many=3), key=PRNGKey(7))
print(P.shape, float(np.nansum(P)), np.array_equal(P, Q, equal_nan=True), float(np.nansum(R_)))
paths = list(scene.all_paths(path_cls=FermatPath, path_cls_kwargs=dict(steps=50), max_order=1, key=PRNGKey(5), approx=True))
print(len(paths), paths[1][3].xys.tolist())
s2 = Scene.random_uniform_scene(n_walls=5, key=PRNGKey(1234))
print(s2.objects[0].xys.tolist())
