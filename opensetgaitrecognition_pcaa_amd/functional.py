"""placeholder -- replaced below"""
