"""subpackage"""
