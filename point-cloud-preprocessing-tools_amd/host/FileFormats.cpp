#include <zlib.h>
#include "FileFormats.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <new>
#include <sstream>

namespace bevio {

namespace {

enum Target { T_NONE = -1, T_X, T_Y, T_Z, T_INTENSITY, T_ROW, T_COL, T_T, T_LABEL };

struct Field {
    std::string name;
    int size = 4;
    char type = 'F';
    int count = 1;
    int offset = 0; /* byte offset inside one packed record */
    Target target = T_NONE; /* which member of PointXYZIRCT the field feeds (resolved once per file) */
};

Target target_of(const std::string &name)
{
    static const char *const names[] = {"x", "y", "z", "intensity", "row", "col", "t", "label"};
    for (int k = 0; k < 8; ++k)
        if (name == names[k]) return (Target)k;
    return T_NONE;
}

double read_scalar(const unsigned char *p, int size, char type)
{
    switch (type) {
    case 'F':
        if (size == 4) { float v; std::memcpy(&v, p, 4); return v; }
        if (size == 8) { double v; std::memcpy(&v, p, 8); return v; }
        break;
    case 'U':
        if (size == 1) return *p;
        if (size == 2) { std::uint16_t v; std::memcpy(&v, p, 2); return v; }
        if (size == 4) { std::uint32_t v; std::memcpy(&v, p, 4); return v; }
        if (size == 8) { std::uint64_t v; std::memcpy(&v, p, 8); return (double)v; }
        break;
    case 'I':
        if (size == 1) return (std::int8_t)*p;
        if (size == 2) { std::int16_t v; std::memcpy(&v, p, 2); return v; }
        if (size == 4) { std::int32_t v; std::memcpy(&v, p, 4); return v; }
        if (size == 8) { std::int64_t v; std::memcpy(&v, p, 8); return (double)v; }
        break;
    }
    return 0.0;
}

/* double -> integer field without undefined behaviour: NaN -> 0, out of range saturates (values a well-formed file of
 * this point type holds are converted exactly) */
template <class T>
T to_int(double v)
{
    if (!(v == v)) return 0;
    const double lo = (double)std::numeric_limits<T>::min(), hi = (double)std::numeric_limits<T>::max();
    return v <= lo ? std::numeric_limits<T>::min() : (v >= hi ? std::numeric_limits<T>::max() : (T)v);
}

void assign_field(pcl::PointXYZIRCT &pt, Target target, double v)
{
    switch (target) {
    case T_X: pt.x = (float)v; break;
    case T_Y: pt.y = (float)v; break;
    case T_Z: pt.z = (float)v; break;
    case T_INTENSITY: pt.intensity = (float)v; break;
    case T_ROW: pt.row = to_int<std::uint16_t>(v); break;
    case T_COL: pt.col = to_int<std::uint16_t>(v); break;
    case T_T: pt.t = to_int<std::uint32_t>(v); break;
    case T_LABEL: pt.label = to_int<std::int16_t>(v); break;
    case T_NONE: break;
    }
}

/* the layout PCL writes for this point type (and savePCDFileBinary below): records are copied member by member */
bool is_native_layout(const std::vector<Field> &fields)
{
    static const int sizes[8] = {4, 4, 4, 4, 2, 2, 4, 2};
    static const char types[8] = {'F', 'F', 'F', 'F', 'U', 'U', 'U', 'I'};
    if (fields.size() != 8) return false;
    for (int k = 0; k < 8; ++k)
        if (fields[k].target != (Target)k || fields[k].size != sizes[k] || fields[k].type != types[k] || fields[k].count != 1)
            return false;
    return true;
}
void unpack_native(pcl::PointXYZIRCT &pt, const unsigned char *p)
{
    std::memcpy(&pt.x, p, 4); std::memcpy(&pt.y, p + 4, 4); std::memcpy(&pt.z, p + 8, 4);
    std::memcpy(&pt.intensity, p + 12, 4); std::memcpy(&pt.row, p + 16, 2); std::memcpy(&pt.col, p + 18, 2);
    std::memcpy(&pt.t, p + 20, 4); std::memcpy(&pt.label, p + 24, 2);
}

/* liblzf decompression (format used by PCL's DATA binary_compressed) */
bool lzf_decompress(const unsigned char *in, std::size_t in_len, unsigned char *out, std::size_t out_len)
{
    const unsigned char *ip = in, *in_end = in + in_len;
    unsigned char *op = out, *out_end = out + out_len;
    while (ip < in_end) {
        unsigned ctrl = *ip++;
        if (ctrl < 32) { /* literal run */
            ++ctrl;
            if (ctrl > (std::size_t)(out_end - op) || ctrl > (std::size_t)(in_end - ip)) return false;
            std::memcpy(op, ip, ctrl);
            op += ctrl; ip += ctrl;
        } else { /* back reference */
            unsigned len = ctrl >> 5;
            if (len == 7) { if (ip >= in_end) return false; len += *ip++; }
            if (ip >= in_end) return false;
            const std::size_t back = ((std::size_t)(ctrl & 0x1f) << 8) + 1 + *ip++;
            len += 2;
            if (back > (std::size_t)(op - out) || len > (std::size_t)(out_end - op)) return false;
            const unsigned char *ref = op - back;
            while (len--) *op++ = *ref++;
        }
    }
    return op == out_end;
}

} // namespace

namespace {
int load_pcd(const std::string &path, pcl::PointCloud<pcl::PointXYZIRCT> &cloud);
}

int loadPCDFile(const std::string &path, pcl::PointCloud<pcl::PointXYZIRCT> &cloud)
{
    try {
        const int rc = load_pcd(path, cloud);
        if (rc != 0) cloud.clear();
        return rc;
    } catch (const std::bad_alloc &) { /* a header that promises more points than memory holds */
        cloud.clear();
        return -1;
    }
}

namespace {
int load_pcd(const std::string &path, pcl::PointCloud<pcl::PointXYZIRCT> &cloud)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return -1;
    f.seekg(0, std::ios::end);
    const std::uint64_t file_size = (std::uint64_t)f.tellg();
    f.seekg(0, std::ios::beg);
    std::vector<Field> fields;
    std::size_t points = 0, width = 0, height = 1;
    bool have_points = false;
    std::string data_kind, line;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ss(line);
        std::string key;
        ss >> key;
        if (key == "FIELDS" || key == "COLUMNS") {
            std::string n;
            while (ss >> n) { Field fd; fd.name = n; fields.push_back(fd); }
        } else if (key == "SIZE") {
            for (auto &fd : fields) ss >> fd.size;
        } else if (key == "TYPE") {
            for (auto &fd : fields) ss >> fd.type;
        } else if (key == "COUNT") {
            for (auto &fd : fields) ss >> fd.count;
        } else if (key == "WIDTH") {
            ss >> width;
        } else if (key == "HEIGHT") {
            ss >> height;
        } else if (key == "POINTS") {
            ss >> points; have_points = true;
        } else if (key == "DATA") {
            ss >> data_kind;
            break;
        }
    }
    if (fields.empty() || fields.size() > 64 || data_kind.empty() || !f) return -1;
    if (!have_points) {
        if (height != 0 && width > std::numeric_limits<std::size_t>::max() / height) return -1;
        points = width * height; /* PCL accepts WIDTH 0 HEIGHT 0 with POINTS n */
    }
    int rec = 0;
    for (auto &fd : fields) {
        if ((fd.size != 1 && fd.size != 2 && fd.size != 4 && fd.size != 8) || fd.count < 1 || fd.count > 4096) return -1;
        if (fd.type != 'F' && fd.type != 'U' && fd.type != 'I') return -1;
        fd.target = target_of(fd.name);
        fd.offset = rec;
        rec += fd.size * fd.count;
    }
    /* the payload cannot hold more points than the bytes that follow the header (ascii: at least "0\n" per value) */
    const std::uint64_t body = file_size - std::min<std::uint64_t>(file_size, (std::uint64_t)f.tellg());
    const std::uint64_t min_bytes_per_point = data_kind == "ascii" ? 2 * fields.size() : (data_kind == "binary" ? (std::uint64_t)rec : 0);
    if (min_bytes_per_point && points > body / min_bytes_per_point) return -1;
    if (data_kind == "binary_compressed" && points > ((std::uint64_t)1 << 32) / (std::uint64_t)rec) return -1;
    cloud.points.assign(points, pcl::PointXYZIRCT{});
    cloud.width = (std::uint32_t)(width ? width : points);
    cloud.height = (std::uint32_t)(height ? height : 1);

    if (data_kind == "ascii") {
        for (std::size_t i = 0; i < points; ++i) {
            if (!std::getline(f, line)) return -1;
            /* token by token with strtod: it accepts "nan" / "inf" like PCL's loader does (operator>> of libstdc++ does
             * not: it fails on them and leaves every later field of the line at 0); a missing or unreadable token
             * gives 0 for that value only */
            const char *cur = line.c_str();
            for (auto &fd : fields)
                for (int k = 0; k < fd.count; ++k) {
                    while (*cur == ' ' || *cur == '\t' || *cur == '\r') ++cur;
                    char *end = nullptr;
                    double v = std::strtod(cur, &end);
                    if (end == cur) { /* not a number: skip the token */
                        v = 0;
                        while (*cur && *cur != ' ' && *cur != '\t' && *cur != '\r') ++cur;
                    } else {
                        cur = end;
                    }
                    if (k == 0) assign_field(cloud.points[i], fd.target, v);
                }
        }
        return 0;
    }
    if (data_kind == "binary") {
        std::vector<unsigned char> buf((std::size_t)rec * points);
        f.read(reinterpret_cast<char *>(buf.data()), (std::streamsize)buf.size());
        if ((std::size_t)f.gcount() != buf.size()) return -1;
        if (is_native_layout(fields)) {
            for (std::size_t i = 0; i < points; ++i) unpack_native(cloud.points[i], buf.data() + i * 26);
            return 0;
        }
        for (std::size_t i = 0; i < points; ++i)
            for (auto &fd : fields)
                if (fd.target != T_NONE)
                    assign_field(cloud.points[i], fd.target, read_scalar(buf.data() + i * rec + fd.offset, fd.size, fd.type));
        return 0;
    }
    if (data_kind == "binary_compressed") {
        std::uint32_t comp = 0, uncomp = 0;
        f.read(reinterpret_cast<char *>(&comp), 4);
        f.read(reinterpret_cast<char *>(&uncomp), 4);
        if (!f || uncomp != (std::uint64_t)rec * points || comp > body) return -1;
        std::vector<unsigned char> cbuf(comp), buf(uncomp);
        f.read(reinterpret_cast<char *>(cbuf.data()), comp);
        if ((std::size_t)f.gcount() != comp) return -1;
        if (!lzf_decompress(cbuf.data(), comp, buf.data(), uncomp)) return -1;
        /* compressed PCDs are stored field-by-field (SoA) */
        std::size_t base = 0;
        for (auto &fd : fields) {
            const std::size_t stride = (std::size_t)fd.size * fd.count;
            if (fd.target != T_NONE)
                for (std::size_t i = 0; i < points; ++i)
                    assign_field(cloud.points[i], fd.target, read_scalar(buf.data() + base + i * stride, fd.size, fd.type));
            base += stride * points;
        }
        return 0;
    }
    return -1;
}
} // namespace

int savePCDFileBinary(const std::string &path, const pcl::PointCloud<pcl::PointXYZIRCT> &cloud)
{
    const std::size_t n = cloud.points.size();
    std::ostringstream h;
    h << "# .PCD v0.7 - Point Cloud Data file format\n"
      << "VERSION 0.7\n"
      << "FIELDS x y z intensity row col t label\n"
      << "SIZE 4 4 4 4 2 2 4 2\n"
      << "TYPE F F F F U U U I\n"
      << "COUNT 1 1 1 1 1 1 1 1\n"
      << "WIDTH " << n << "\n"
      << "HEIGHT 1\n"
      << "VIEWPOINT 0 0 0 1 0 0 0\n"
      << "POINTS " << n << "\n"
      << "DATA binary\n";
    const std::string head = h.str();
    std::vector<unsigned char> buf(head.size() + n * 26);
    std::memcpy(buf.data(), head.data(), head.size());
    unsigned char *p = buf.data() + head.size();
    for (const auto &pt : cloud.points) { /* packed 26-byte records */
        std::memcpy(p, &pt.x, 4); std::memcpy(p + 4, &pt.y, 4); std::memcpy(p + 8, &pt.z, 4);
        std::memcpy(p + 12, &pt.intensity, 4); std::memcpy(p + 16, &pt.row, 2); std::memcpy(p + 18, &pt.col, 2);
        std::memcpy(p + 20, &pt.t, 4); std::memcpy(p + 24, &pt.label, 2);
        p += 26;
    }
    return writeFile(path, buf.data(), buf.size()) ? 0 : -1;
}

bool writeFile(const std::string &path, const void *data, std::size_t n)
{
    std::FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(data, 1, n, f) == n;
    return std::fclose(f) == 0 && ok;
}

/* ---- PNG: 8-bit grayscale, filter 0 on every row, one IDAT chunk ---- */
namespace {
std::uint32_t png_crc(const unsigned char *p, std::size_t n) /* zlib's CRC-32 is the PNG chunk CRC; no shared state */
{
    return (std::uint32_t)::crc32(::crc32(0L, Z_NULL, 0), p, (uInt)n);
}
void put32(std::vector<unsigned char> &v, std::uint32_t x)
{
    v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x);
}
void chunk(std::vector<unsigned char> &out, const char *tag, const std::vector<unsigned char> &body)
{
    put32(out, (std::uint32_t)body.size());
    std::vector<unsigned char> t(tag, tag + 4);
    t.insert(t.end(), body.begin(), body.end());
    out.insert(out.end(), t.begin(), t.end());
    put32(out, png_crc(t.data(), t.size()));
}
} // namespace

bool writePngGray8(const std::string &path, const std::uint8_t *pixels, int rows, int cols)
{
    std::vector<unsigned char> raw;
    raw.reserve((std::size_t)rows * (cols + 1));
    for (int r = 0; r < rows; ++r) {
        raw.push_back(0); /* filter: none */
        raw.insert(raw.end(), pixels + (std::size_t)r * cols, pixels + (std::size_t)(r + 1) * cols);
    }
    /* zlib stream with real deflate (the system's libz; level 1 like cv::imwrite's default: the sparse occupancy
     * layers shrink ~50x, which is what the CLI's disk time is made of) */
    uLongf zn = compressBound((uLong)raw.size());
    std::vector<unsigned char> z(zn);
    if (compress2(z.data(), &zn, raw.data(), (uLong)raw.size(), 1) != Z_OK) return false;
    z.resize(zn);
    std::vector<unsigned char> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<unsigned char> ihdr;
    put32(ihdr, (std::uint32_t)cols); put32(ihdr, (std::uint32_t)rows);
    ihdr.push_back(8); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr);
    chunk(out, "IDAT", z);
    chunk(out, "IEND", {});
    return writeFile(path, out.data(), out.size());
}

/* cv::format(mat, cv::Formatter::FMT_CSV) for CV_8U, from memory of OpenCV's out.cpp:
 * every value "%3d", values joined by ", ", every row terminated by "\n". */
std::string formatCsvU8(const std::uint8_t *pixels, int rows, int cols)
{
    std::string s;
    s.reserve((std::size_t)rows * (cols * 5));
    char buf[8];
    for (int r = 0; r < rows; ++r) {
        for (int c = 0; c < cols; ++c) {
            std::snprintf(buf, sizeof buf, "%3d", (int)pixels[(std::size_t)r * cols + c]);
            s += buf;
            if (c + 1 < cols) s += ", ";
        }
        s += "\n";
    }
    return s;
}

} // namespace bevio
