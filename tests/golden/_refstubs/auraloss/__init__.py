"""empty stub"""
