from .knn import knn_points
