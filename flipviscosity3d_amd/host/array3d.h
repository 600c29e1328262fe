// array3d.h -- host-side dense 3-D container with the reference's memory layout.
//
// Mirrors the contract of the reference's Array3d<T> (reference array3d.h:61-420) that the rest of a
// frame loop relies on: public width/height/depth, flat index i + width*(j + height*k) (:397-400),
// getRawArray() as the zero-copy handle (:341-343), bounds-checked accessors that throw
// std::out_of_range unless an out-of-range value is set (:140-153, :349-355), negative dimensions ->
// std::domain_error (:378-384).  Storage is a std::vector (value semantics, no manual new/delete).
#pragma once

#include <sstream>
#include <stdexcept>
#include <vector>

struct GridIndex {  // reference array3d.h:33-59
    int i, j, k;
    GridIndex() : i(0), j(0), k(0) {}
    GridIndex(int ii, int jj, int kk) : i(ii), j(jj), k(kk) {}
    bool operator==(const GridIndex &o) const { return i == o.i && j == o.j && k == o.k; }
    bool operator!=(const GridIndex &o) const { return !(*this == o); }
    int operator[](int idx) const { return idx == 0 ? i : (idx == 1 ? j : k); }
};

template <class T>
class Array3d {
public:
    int width = 0, height = 0, depth = 0;

    Array3d() {}
    Array3d(int i, int j, int k) { _init(i, j, k); }
    Array3d(int i, int j, int k, T fillValue) {
        _init(i, j, k);
        fill(fillValue);
    }

    void fill(T value) { std::fill(_grid.begin(), _grid.end(), value); }

    T operator()(int i, int j, int k) const { return get(i, j, k); }
    T operator()(GridIndex g) const { return get(g.i, g.j, g.k); }
    T get(int i, int j, int k) const {
        if (!isIndexInRange(i, j, k)) {
            if (_hasOutOfRange) return _outOfRange;
            _throwRange(i, j, k);
        }
        return _grid[flatIndex(i, j, k)];
    }
    T get(GridIndex g) const { return get(g.i, g.j, g.k); }

    void set(int i, int j, int k, T value) {
        if (!isIndexInRange(i, j, k)) _throwRange(i, j, k);
        _grid[flatIndex(i, j, k)] = value;
    }
    void set(GridIndex g, T value) { set(g.i, g.j, g.k, value); }
    void add(int i, int j, int k, T value) {
        if (!isIndexInRange(i, j, k)) _throwRange(i, j, k);
        _grid[flatIndex(i, j, k)] += value;
    }

    T *getPointer(int i, int j, int k) {
        if (!isIndexInRange(i, j, k)) _throwRange(i, j, k);
        return &_grid[flatIndex(i, j, k)];
    }
    T *getRawArray() { return _grid.data(); }
    const T *getRawArray() const { return _grid.data(); }
    int getNumElements() const { return (int)_grid.size(); }
    size_t size() const { return _grid.size(); }

    void setOutOfRangeValue() { _hasOutOfRange = false; }
    void setOutOfRangeValue(T v) {
        _outOfRange = v;
        _hasOutOfRange = true;
    }
    bool isOutOfRangeValueSet() const { return _hasOutOfRange; }
    T getOutOfRangeValue() const { return _outOfRange; }

    bool isIndexInRange(int i, int j, int k) const {
        return i >= 0 && j >= 0 && k >= 0 && i < width && j < height && k < depth;
    }
    bool isIndexInRange(GridIndex g) const { return isIndexInRange(g.i, g.j, g.k); }

    size_t flatIndex(int i, int j, int k) const {
        return (size_t)i + (size_t)width * ((size_t)j + (size_t)height * (size_t)k);
    }

private:
    void _init(int i, int j, int k) {
        if (i < 0 || j < 0 || k < 0) {
            std::ostringstream s;
            s << "Error: dimensions cannot be negative.\nwidth: " << i << " height: " << j << " depth: " << k << "\n";
            throw std::domain_error(s.str());
        }
        width = i; height = j; depth = k;
        _grid.assign((size_t)i * j * k, T());
    }
    [[noreturn]] void _throwRange(int i, int j, int k) const {
        std::ostringstream s;
        s << "Error: index out of range.\ni: " << i << " j: " << j << " k: " << k << "\n";
        throw std::out_of_range(s.str());
    }

    std::vector<T> _grid;
    bool _hasOutOfRange = false;
    T _outOfRange = T();
};
