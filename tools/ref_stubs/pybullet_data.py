def getDataPath():
    return "/nonexistent/pybullet_data"
